"""Block-Jacobi vs multigrid PCG across Manhattan graph sizes (where should the automatic rule switch?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sim3opt_amd import lib as L, synth
synth.DRIFT_TARGET = 0.05
for V in (1000, 2000, 4000, 8000, 20000, 50000):
    side = max(4, int(round((V / 10.0) ** 0.5)))
    g = synth.manhattan(V, 10 * V, dims=(side, side, 10), per_cell=4)
    for pre in (0, 2):
        G = L.Graph(fix_small_angle_b=1, pcg_rel_tol=1e-8, preconditioner=pre)
        G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"]); G.initialize()
        G.optimize(1); G.set_vertices(g["states"])
        t = time.perf_counter(); G.optimize(10); dt = time.perf_counter() - t
        st = G.stats()
        print("V %6d pre %d (in use %d): 10 LM it %.3fs chi %.6g pcg %s" % (V, pre, G.preconditioner_in_use(), dt, st[-1].chi2_after, [s.pcg_iters for s in st]), flush=True)
        G.close()
