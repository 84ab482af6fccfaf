"""KITTI-00 direct PGO (exact-B arithmetic), 100 LM iterations: PCG preconditioners side by side."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from sim3opt_amd import lib as L, synth
import kitti_graph as K
for one in (True, False):
    g = K.build_direct_graph(one)
    ref = None
    for pre, env in ((0, {}), (1, {}), (2, {}), (2, {"SIM3OPT_AMG_ADDITIVE": "0"})):
        for k in ("SIM3OPT_AMG_ADDITIVE",):
            os.environ.pop(k, None)
        os.environ.update(env)
        G = L.Graph(fix_small_angle_b=1, preconditioner=pre, pcg_rel_tol=1e-10, pcg_max_iters=40000)
        G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"]); G.initialize()
        G.optimize(1); G.set_vertices(g["states"])
        t = time.perf_counter(); n = G.optimize(100); dt = time.perf_counter() - t
        st = G.stats()
        x = G.get_vertices()
        if ref is None:
            ref = x
        print("one_loop %s pre %d %s: %d LM it %.3fs chi %.8g pcg total %d (max %d) rmse vs pre0 %.2e" % (
            one, pre, env, n, dt, st[-1].chi2_after, sum(s.pcg_iters for s in st), max(s.pcg_iters for s in st), synth.rmse(x, ref)), flush=True)
        G.close()
