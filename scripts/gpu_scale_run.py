"""Scale check: 1M vertices / 10M edges Manhattan graph on one MI355X (8.2 GB of 7x7 blocks,
5.6 GB of assembly scratch) -- same code path as the 100k/1M bench."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sim3opt_amd import lib as L, synth
V = int(os.environ.get("V", "1000000")); E = 10 * V
synth.DRIFT_TARGET = 0.05
t = time.time(); g = synth.manhattan(V, E, dims=(int(round((V / 10) ** 0.5)),) * 2 + (10,)); print("generated in %.1fs" % (time.time() - t), flush=True)
PRE = int(os.environ.get("PRE", "-1"))
G = L.Graph(fix_small_angle_b=1, pcg_rel_tol=1e-8, time_kernels=1, preconditioner=PRE, verbose=1)
t = time.time(); G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"]); G.initialize(); print("host build + upload %.1fs" % (time.time() - t), flush=True)
nb, nnzb = G.system_dims()
chi0 = G.chi2()
t = time.time(); n = G.optimize(int(os.environ.get("NIT", "4"))); dt = time.time() - t
st = G.stats(); kt = G.kernel_times()
byt = nnzb * 396 + (nb + 1) * 4 + 3 * 7 * nb * 8
out = dict(vertices=V, edges=E, preconditioner=G.preconditioner_in_use(), pcg_rel_res=[s.pcg_rel_res for s in st], ms_solve=[s.ms_solve for s in st], blocks=nnzb, vals_GB=nnzb * 392 / 1e9, lm_iters=n, seconds=dt, lm_iters_per_s=n / dt, edges_iters_per_s=E * n / dt,
           chi2=[chi0] + [s.chi2_after for s in st], pcg_iters=[s.pcg_iters for s in st], ms_linearize=[s.ms_linearize for s in st],
           block_array_GB=G.device_bytes()[0] / 1e9,
           spmv_ms=kt.ms_spmv / max(kt.n_spmv, 1), spmv_GBs=byt / (kt.ms_spmv / max(kt.n_spmv, 1)) / 1e6, b2b_spmv_ms=G.bench_spmv(20))
print(json.dumps(out))
outdir = os.environ.get("OUT", "") if os.path.isdir(os.environ.get("OUT", "")) else os.path.join(ROOT, "gpurun_out")
os.makedirs(outdir, exist_ok=True)
json.dump(out, open(os.path.join(outdir, os.environ.get("OUTFILE", "scale_V%d_pre%d.json" % (V, G.preconditioner_in_use()))), "w"), indent=1)
