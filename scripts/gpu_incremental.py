"""BASELINE.json configs[4] in its own words: KITTI-00 "incremental loop-closure" Sim(3) PGO -- the
loop constraints of loopConstraints.txt are added one at a time, and after each one LM (at most 100
iterations, g2o's stopping rule) runs warm-started from the previous solution.  Reference
configuration (delta = 1e-9, B as written).  GPU (libsim3opt through the C-ABI: graph growth +
sim3opt_initialize keeps the estimates) next to the CPU oracle doing the same.
Writes gpurun_out/r4_incremental_fixb<0|1>.json.
Usage: python scripts/gpu_incremental.py [max_iters=100] [--exact-b] [--no-cpu]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from sim3opt_amd import lib as L, synth  # noqa: E402
import kitti_graph as K  # noqa: E402

MAXIT = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 100
FIXB = 1 if "--exact-b" in sys.argv else 0  # 0: sim3_rv.h's small-angle coefficient as written (reference)
full = K.build_direct_graph(False)
nl = 118  # the loop edges come first in the builder's edge list, then the 770 odometry edges
gt = np.loadtxt(os.path.join(K.FIXTURE, "gt_kf.txt"), comments="%")[:, [4, 8, 12]]


G = L.Graph(fix_small_angle_b=FIXB)
G.add_vertices(full["states"], full["fixed"])
G.add_edges(full["v0"][nl:], full["v1"][nl:], full["meas"][nl:])
per = []
t0 = time.perf_counter()
for k in range(nl):
    G.add_edge(int(full["v0"][k]), int(full["v1"][k]), full["meas"][k])
    t = time.perf_counter()
    G.initialize()
    t_init = time.perf_counter() - t
    n = max(G.optimize(MAXIT), 0)
    st = G.stats()
    per.append(dict(closure=k, iters=n, ms=1e3 * (time.perf_counter() - t), ms_initialize=1e3 * t_init,
                    chi2=st[-1].chi2_after if st else G.chi2(), solver=G.linear_solver_in_use()))
t_gpu = time.perf_counter() - t0
out = dict(fix_small_angle_b=FIXB, max_iters_per_closure=MAXIT, closures=nl, gpu_total_seconds=t_gpu,
           gpu_lm_iterations=int(sum(p["iters"] for p in per)),
           gpu_mean_ms_per_closure=float(np.mean([p["ms"] for p in per])),
           gpu_median_ms_initialize=float(np.median([p["ms_initialize"] for p in per])),
           gpu_first_ms_initialize=per[0]["ms_initialize"],
           gpu_final_chi2=per[-1]["chi2"], gpu_exact_solver_closures=int(sum(p["solver"] == 1 for p in per)),
           gpu_rmse_vs_gt_m=L.align_trajectory(synth.positions(G.get_vertices()), gt)[1],
           gpu_iters_per_closure=[p["iters"] for p in per], gpu_first=per[:3], gpu_last=per[-3:])
print(json.dumps({k: v for k, v in out.items() if not k.endswith(("first", "last", "per_closure"))}), flush=True)
if "--no-cpu" not in sys.argv:
    from oracle import oracle as O  # checker / baseline only
    states = full["states"].copy()
    cper = []
    t0 = time.perf_counter()
    for k in range(nl):
        # (same edge order as the GPU graph: odometry first, then the loops added so far)
        idx = np.r_[np.arange(nl, len(full["v0"])), np.arange(k + 1)]
        t = time.perf_counter()
        OG = O.Graph(states, full["fixed"], full["v0"][idx], full["v1"][idx], full["meas"][idx])
        it, tr = OG.optimize(MAXIT, O.default_options(fix_small_angle_b=FIXB))
        states = OG.states.copy()
        cper.append(dict(closure=k, iters=max(it, 0), ms=1e3 * (time.perf_counter() - t),
                         chi2=tr[-1].chi2_after if tr else float("nan")))
    t_cpu = time.perf_counter() - t0
    out.update(cpu_total_seconds=t_cpu, cpu_cores=1, cpu_lm_iterations=int(sum(p["iters"] for p in cper)),
               cpu_mean_ms_per_closure=float(np.mean([p["ms"] for p in cper])), cpu_final_chi2=cper[-1]["chi2"],
               cpu_rmse_vs_gt_m=L.align_trajectory(synth.positions(states), gt)[1],
               rmse_gpu_vs_cpu_m=synth.rmse(G.get_vertices(), states), speedup=t_cpu / t_gpu,
               closures_with_equal_iteration_count=int(sum(a["iters"] == b["iters"] for a, b in zip(per, cper))),
               cpu_iters_per_closure=[p["iters"] for p in cper], cpu_first=cper[:3], cpu_last=cper[-3:])
    print(json.dumps({k: v for k, v in out.items() if k.startswith(("cpu_", "rmse", "speed", "closures_w")) and not k.endswith(("first", "last"))}), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r4_incremental_fixb%d.json" % FIXB), "w"), indent=1)
