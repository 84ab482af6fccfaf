"""Robustness sweep of the automatic (multigrid) path: seeds, drift, robust kernel, dense information."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sim3opt_amd import lib as L, synth
V, E = 20000, 200000
for seed in (1, 2, 3):
    for drift in (0.05, 0.3):
        synth.DRIFT_TARGET = drift
        g = synth.manhattan(V, E, dims=(45, 45, 10), seed_graph=1000 + seed, seed_noise=2000 + seed)
        for variant in ("plain", "huber", "info"):
            kw = {}
            if variant == "huber":
                kw = dict(kernel=L.KERNEL_HUBER, kernel_delta=1.0)
            if variant == "info":
                rng = np.random.default_rng(seed)
                M = rng.standard_normal((E, 7, 7)) * 0.3
                kw = dict(info=np.einsum("kij,klj->kil", M, M) + np.eye(7))
            G = L.Graph(fix_small_angle_b=1, pcg_rel_tol=1e-8, verbose=0)
            G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"], **kw); G.initialize()
            t = time.perf_counter(); n = G.optimize(15); dt = time.perf_counter() - t
            st = G.stats()
            bad = [i for i, s in enumerate(st) if s.pcg_rel_res > 1e-8 and s.trials == 1]
            print("seed %d drift %.2f %-5s pre %d: %2d it %.2fs chi %.5g -> %.5g pcg max %d trials max %d unconverged %s rmse-to-gt %.3f -> %.3f" % (
                seed, drift, variant, G.preconditioner_in_use(), n, dt, st[0].chi2_before, st[-1].chi2_after, max(s.pcg_iters for s in st),
                max(s.trials for s in st), bad, synth.rmse(g["states"], g["gt"]), synth.rmse(G.get_vertices(), g["gt"])), flush=True)
            G.close()
