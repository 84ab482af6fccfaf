"""Per-LM-iteration breakdown of the driver's window on config 3 (5 warm-up + 20 timed iterations):
trials, PCG iterations, ms in linearisation / solve / update, preconditioner in use."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sim3opt_amd import lib as L, synth
synth.DRIFT_TARGET = 0.05
g = synth.manhattan(100000, 1000000)
G = L.Graph(fix_small_angle_b=1, pcg_rel_tol=1e-8, time_kernels=0)
G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"]); G.initialize()
G.optimize(5); G.set_vertices(g["states"])
t = time.perf_counter(); G.optimize(25); dt = time.perf_counter() - t
st = G.stats()
print("25 iterations %.1f ms" % (dt * 1e3))
for i, s in enumerate(st):
    print("it %2d  trials %d  pcg %4d  lin %.2f  solve %6.2f  upd %.2f  lambda %.3g  chi2 %.6g" % (
        i + 1, s.trials, s.pcg_iters, s.ms_linearize, s.ms_solve, s.ms_update, s.lambda_, s.chi2_after))
