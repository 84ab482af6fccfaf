"""What an N-rank partition replicates, measured on ONE GPU: the hierarchy and cycle an N-rank run builds
(options.row_order = 1, amg_virtual_ranks = N), device time of every visit of the first replicated multigrid
level bracketed by HIP events (sim3opt_kernel_times.ms_replicated_levels) next to the whole solve time -- the
two numbers DESIGN.md 7's model is built from.
Usage: python scripts/gpu_replicated_share.py [N=8] [V=100000 E=1000000] [iters=8] [cycle digits, e.g. 12]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sim3opt_amd import lib as L, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
V = int(sys.argv[2]) if len(sys.argv) > 3 else 100000
E = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
ITERS = int(sys.argv[4]) if len(sys.argv) > 4 else 8
CYC = sys.argv[5] if len(sys.argv) > 5 else ""
OUT = os.environ.get("OUT", os.path.join(ROOT, "gpurun_out"))
synth.DRIFT_TARGET = 0.05
g = synth.manhattan(V, E) if V == 100000 else synth.manhattan(V, E, dims=(int(round((V / 10) ** 0.5)),) * 2 + (10,))
kw = {}
if CYC:
    kw["amg_cycle"] = [int(c) for c in (CYC + CYC[-1] * 4)[:4]]
G = L.Graph(fix_small_angle_b=1, pcg_rel_tol=1e-8, time_kernels=1, row_order=1, amg_virtual_ranks=N, **kw)
G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"])
G.initialize()
G.optimize(2); G.set_vertices(g["states"]); G.kernel_times(reset=True)
n = G.optimize(ITERS)
st = G.stats(); kt = G.kernel_times()
npcg = sum(int(s.pcg_iters) for s in st)
rows, blocks, _ = G.amg_hierarchy()
o = G.options()
out = dict(virtual_ranks=N, vertices=V, edges=E, lm_iters=int(n), pcg_iters=[int(s.pcg_iters) for s in st],
           levels_rows=[int(x) for x in rows], levels_blocks=[int(x) for x in blocks], amg_cycle_option=list(o.amg_cycle),
           amg_shard_rows=int(o.amg_shard_rows),
           ms_solve=[float(s.ms_solve) for s in st],
           ms_solve_per_pcg_iteration=float(sum(s.ms_solve for s in st)) / max(1, npcg),
           ms_replicated_levels_per_pcg_iteration=kt.ms_replicated_levels / max(1, npcg),
           replicated_visits_per_pcg_iteration=kt.n_replicated_visits / max(1, npcg),
           ms_spmv_per_launch=kt.ms_spmv / max(1, kt.n_spmv),
           ms_linearize_mean=float(np.mean([s.ms_linearize for s in st])), chi2=[float(s.chi2_after) for s in st])
out["replicated_share_of_solve"] = out["ms_replicated_levels_per_pcg_iteration"] / out["ms_solve_per_pcg_iteration"]
print(json.dumps(out))
tag = ("" if V == 100000 else "_V%d" % V) + ("_cycle%s" % CYC if CYC else "")
json.dump(out, open(os.path.join(OUT, "r4_replicated_share_N%d%s.json" % (N, tag)), "w"), indent=1)
