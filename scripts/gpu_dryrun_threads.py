"""Dry run of the row-partitioned path with N ranks as THREADS of one process on one GPU (a box admits six
processes on its card; threads are not counted): the real kernels, the partition-aware hierarchy, neighbour
exchanges and the collective sequence, host collectives through tests/dist_helpers.ThreadGroup.  The times mean
nothing (N ranks share one GPU); payloads, collective counts and the partition are exact.
Usage: python scripts/gpu_dryrun_threads.py [N=8] [V=100000 E=1000000] [iters=4]
   -> $OUT/r4_dryrun_threads_N<N>[_V<V>].json   (OUT defaults to <repo>/gpurun_out)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import dist_helpers as H
from sim3opt_amd import lib as L, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
V = int(sys.argv[2]) if len(sys.argv) > 3 else 100000
E = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
ITERS = int(sys.argv[4]) if len(sys.argv) > 4 else 4
OUT = os.environ.get("OUT", os.path.join(ROOT, "gpurun_out"))
synth.DRIFT_TARGET = 0.05
g = synth.manhattan(V, E) if V == 100000 else synth.manhattan(V, E, dims=(int(round((V / 10) ** 0.5)),) * 2 + (10,))
tg = H.ThreadGroup(N, timeout=1800.0)

def body(rank):
    G = L.Graph(device=0, fix_small_angle_b=1, pcg_rel_tol=1e-8, time_kernels=1)
    G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"])
    tg.attach(G, rank)
    G.initialize()
    G.kernel_times(reset=True)
    n = G.optimize(ITERS)
    st = G.stats(); ct = G.comm_times()
    lo, hi = G.local_rows()
    out = dict(rank=rank, rows=[int(lo), int(hi)], lm_iters=int(n), pcg_iters=[int(s.pcg_iters) for s in st],
               trials=[int(s.trials) for s in st], chi2=[float(s.chi2_after) for s in st],
               comm={k: (float(v) if isinstance(v, float) else int(v)) for k, v in ct.items()},
               block_array_bytes=list(G.device_bytes()), debug_full_arrays=int(G.options().debug_full_arrays))
    if rank == 0:
        _, _, bnd, cut = G.partition_plan(N)
        o = G.options()
        out.update(boundary_rows_per_rank=[int(x) for x in bnd], cut_edges=int(cut),
                   amg_cycle_used=list(o.amg_cycle), amg_shard_rows=int(o.amg_shard_rows))
    G.close()
    return out

res = tg.run(body)
r0 = res[0]
npcg = max(1, sum(r0["pcg_iters"]))
nlm = max(1, r0["lm_iters"])
def per(rank_rec):
    c = rank_rec["comm"]
    return dict(exchange_MB_per_pcg_iteration=c["bytes_exchange"] / npcg / 1e6,
                allgather_MB_per_pcg_iteration=c["bytes_allgather"] / npcg / 1e6,
                allreduce_MB_per_lm_iteration=c["bytes_allreduce"] / nlm / 1e6,
                exchanges_per_pcg_iteration=c["n_exchange"] / npcg, allgathers_per_pcg_iteration=c["n_allgather"] / npcg,
                allreduces_per_pcg_iteration=c["n_allreduce"] / npcg)
# the single-rank reference with the same row order, hierarchy and cycle
R = L.Graph(device=0, fix_small_angle_b=1, pcg_rel_tol=1e-8, row_order=1, amg_virtual_ranks=N)
R.add_vertices(g["states"], g["fixed"]); R.add_edges(g["v0"], g["v1"], g["meas"])
R.initialize()
R.optimize(ITERS)
rs = R.stats()
rows, blocks, _ = R.amg_hierarchy()
summary = dict(ranks=N, vertices=V, edges=E, transport="threads of one process, host-staged", lm_iters=r0["lm_iters"],
               pcg_iters=r0["pcg_iters"], trials=r0["trials"], chi2=r0["chi2"],
               one_rank_same_hierarchy=dict(pcg_iters=[int(s.pcg_iters) for s in rs], trials=[int(s.trials) for s in rs],
                                            chi2=[float(s.chi2_after) for s in rs]),
               max_rel_chi2_diff_to_one_rank=float(max(abs(a - s.chi2_after) / s.chi2_after for a, s in zip(r0["chi2"], rs))),
               levels_rows=[int(x) for x in rows], levels_blocks=[int(x) for x in blocks],
               boundary_rows_per_rank=r0["boundary_rows_per_rank"], cut_edges=r0["cut_edges"],
               amg_cycle_option=r0["amg_cycle_used"], amg_shard_rows=r0["amg_shard_rows"],
               per_rank=[per(r) for r in res],
               block_array_MB_per_rank=[round(r["block_array_bytes"][0] / 1e6, 1) for r in res],
               block_array_MB_one_rank=round(r0["block_array_bytes"][1] / 1e6, 1), debug_full_arrays=r0["debug_full_arrays"],
               identical_chi2_on_all_ranks=all(r["chi2"] == r0["chi2"] for r in res), rank0=r0)
print(json.dumps({k: v for k, v in summary.items() if k not in ("rank0", "per_rank")}))
print(json.dumps(dict(rank0=per(res[0]), worst_exchange_MB_per_pcg_iteration=max(p["exchange_MB_per_pcg_iteration"] for p in summary["per_rank"]))))
tag = "" if V == 100000 else "_V%d" % V
json.dump(summary, open(os.path.join(OUT, "r4_dryrun_threads_N%d%s.json" % (N, tag)), "w"), indent=1)
