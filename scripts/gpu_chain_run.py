import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle as O
from sim3opt_amd import lib as L, synth
import kitti_graph as K
for one in (True, False):
    g = K.build_direct_graph(one)
    OG = O.Graph(g['states'], g['fixed'], g['v0'], g['v1'], g['meas'])
    o = O.default_options(fix_small_angle_b=1, fd_delta=1e-6)
    H, b = OG.build_dense(o)
    lam = 1e-5 * np.abs(np.diag(H)).max()
    xd = np.linalg.solve(H + lam * np.eye(len(b)), b)
    for pre, seg in ((0, 256), (1, 64), (1, 256)):
        G = L.Graph(fix_small_angle_b=1, fd_delta=1e-6, pcg_rel_tol=1e-12, pcg_max_iters=40000, preconditioner=pre, chain_segment=seg)
        G.add_vertices(g['states'], g['fixed']); G.add_edges(g['v0'], g['v1'], g['meas']); G.initialize(); G.linearize()
        t = time.perf_counter(); x, it, rr = G.solve(lam); dt = time.perf_counter() - t
        print("one", one, "pre", pre, "seg", seg, "iters", it, "relres %.1e" % rr, "ms %.2f" % (1e3 * dt), "x err %.1e" % (np.abs(x - xd).max() / np.abs(xd).max()), flush=True)
        t = time.perf_counter(); n = G.optimize(30); dt = time.perf_counter() - t
        st = G.stats()
        print("     optimize 30: %.2fs" % dt, "chi %.6f" % st[-1].chi2_after, "pcg total", sum(s.pcg_iters for s in st))
synth.DRIFT_TARGET = 0.05
g = synth.chain_loop(10000, 20000)
for pre in (0, 1):
    G = L.Graph(fix_small_angle_b=1, pcg_rel_tol=1e-8, preconditioner=pre)
    G.add_vertices(g['states'], g['fixed']); G.add_edges(g['v0'], g['v1'], g['meas']); G.initialize()
    t = time.perf_counter(); n = G.optimize(8); dt = time.perf_counter() - t
    st = G.stats(); print("chain_loop 10k/20k pre", pre, "%.2fs" % dt, "chi", ["%.4g" % s.chi2_after for s in st], "pcg", [s.pcg_iters for s in st])
