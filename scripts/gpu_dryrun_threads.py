"""Dry run of the row-partitioned path with N ranks as THREADS of one process on one GPU (a box admits six
processes on its card; threads are not counted): config 3, 4 LM iterations, the real kernels, halo exchange and
collective sequence, host collectives through tests/dist_helpers.ThreadGroup.  The times mean nothing (N ranks
share one GPU); the payloads per PCG iteration and the partition are exact.
Usage: python scripts/gpu_dryrun_threads.py [N=8]   -> gpurun_out/r3_dryrun_threads_N<N>.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import dist_helpers as H
from sim3opt_amd import lib as L, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
synth.DRIFT_TARGET = 0.05
g = synth.manhattan(100000, 1000000)
tg = H.ThreadGroup(N)

def body(rank):
    G = L.Graph(device=0, fix_small_angle_b=1, pcg_rel_tol=1e-8, time_kernels=1)
    G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"])
    G.comm_init_callbacks(rank, N, tg.allreduce(rank), tg.allgatherv(rank))
    G.initialize()
    G.kernel_times(reset=True)
    n = G.optimize(4)
    st = G.stats(); ct = G.comm_times()
    lo, hi = G.local_rows()
    _, _, bnd, cut = G.partition_plan(N)
    out = dict(rank=rank, rows=[int(lo), int(hi)], lm_iters=int(n), pcg_iters=[int(s.pcg_iters) for s in st],
               chi2=[float(s.chi2_after) for s in st], comm={k: (float(v) if isinstance(v, float) else int(v)) for k, v in ct.items()},
               boundary_rows_all_ranks=int(bnd.sum()), cut_edges=int(cut))
    G.close()
    return out

res = tg.run(body)
r0 = res[0]
npcg = max(1, sum(r0["pcg_iters"]))
summary = dict(ranks=N, transport="threads of one process, host-staged", lm_iters=r0["lm_iters"], pcg_iters=r0["pcg_iters"],
               chi2=r0["chi2"], boundary_rows_all_ranks=r0["boundary_rows_all_ranks"], cut_edges=r0["cut_edges"],
               allgather_MB_per_pcg_iteration=r0["comm"].get("bytes_allgather", 0) / npcg / 1e6,
               allreduce_MB_per_lm_iteration=r0["comm"].get("bytes_allreduce", 0) / max(1, r0["lm_iters"]) / 1e6,
               identical_chi2_on_all_ranks=all(r["chi2"] == r0["chi2"] for r in res), rank0=r0)
print(json.dumps({k: v for k, v in summary.items() if k != "rank0"}))
json.dump(summary, open(os.path.join(ROOT, "gpurun_out", "r3_dryrun_threads_N%d.json" % N), "w"), indent=1)
