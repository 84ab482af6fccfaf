import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sim3opt_amd import lib as L, synth
def run(g, tag, iters=8, **opts):
    G = L.Graph(**opts); G.add_vertices(g['states'], g['fixed']); G.add_edges(g['v0'], g['v1'], g['meas']); G.initialize()
    c0 = G.chi2(); t = time.time()
    done = 0; st = []
    while done < iters:
        n = G._L.sim3opt_optimize(G._g, iters - done)
        if n <= 0: break
        done += n; st += G.stats()
    dt = time.time() - t
    print(tag, "chi0 %.4g" % c0, "t %.2fs" % dt, "chi", ["%.4g" % s.chi2_after for s in st], "trials", [s.trials for s in st], "pcg", [s.pcg_iters for s in st], "rel", ["%.1e" % s.pcg_rel_res for s in st], "lam", ["%.1e" % s.lambda_ for s in st], "rmse_gt %.3f (init %.3f)" % (synth.rmse(G.get_vertices(), g['gt']), synth.rmse(g['states'], g['gt'])), flush=True)
for drift in (0.05, 0.01):
    synth.DRIFT_TARGET = drift
    t = time.time(); g = synth.manhattan(); print("gen", time.time() - t, flush=True)
    Ggt = L.Graph(); Ggt.add_vertices(g['gt'], g['fixed']); Ggt.add_edges(g['v0'], g['v1'], g['meas']); Ggt.initialize(); print("drift", drift, "chi2 at gt", Ggt.chi2(), flush=True); del Ggt
    for fixb in (0, 1):
        for cap in (200, 2000):
            run(g, "drift %.2f fixb %d cap %d" % (drift, fixb, cap), fix_small_angle_b=fixb, pcg_max_iters=cap, pcg_rel_tol=1e-8)
