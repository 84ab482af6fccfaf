"""Config 3 the way the reference calls the optimizer: optimize(100) (kitti_surf.cpp:675)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sim3opt_amd import lib as L, synth
synth.DRIFT_TARGET = 0.05
g = synth.manhattan()
out = {}
for pre in (-1, 0):
    G = L.Graph(fix_small_angle_b=1, pcg_rel_tol=1e-8, preconditioner=pre)
    G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"]); G.initialize()
    chi0 = G.chi2()
    t = time.perf_counter(); n = 0; st = []
    while n < 100:
        k = G._L.sim3opt_optimize(G._g, 100 - n)
        if k <= 0: break
        n += k; st += G.stats()
        if st[-1].trials >= 10 or st[-1].rho == 0: break   # g2o Terminate
    dt = time.perf_counter() - t
    rec = dict(preconditioner=G.preconditioner_in_use(), iters=n, seconds=dt, lm_iters_per_s=n / dt, chi2_0=chi0,
               chi2=[s.chi2_after for s in st][:15] + ["..."] + [st[-1].chi2_after],
               pcg_iters_total=int(sum(s.pcg_iters for s in st)), trials_total=int(sum(s.trials for s in st)),
               unconverged_solves=int(sum(1 for s in st if s.pcg_rel_res > 1e-8)),
               rmse_to_gt=[synth.rmse(g["states"], g["gt"]), synth.rmse(G.get_vertices(), g["gt"])])
    out["preconditioner=%d" % pre] = rec
    print(json.dumps(rec), flush=True)
    G.close()
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r2_cfg3_optimize100.json"), "w"), indent=1)
