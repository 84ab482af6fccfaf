"""How well do independent multigrid-PCG solves overlap on one GPU?  T host threads, each with its own
graph (its own HIP stream), run the same 8 LM iterations of config 3 at the same time: wall time against
one thread.  (What speculative solves of the rejected trials of one LM iteration on separate streams
could gain: the coarse cycle is latency-bound, the level-0 passes are bandwidth-bound.)"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sim3opt_amd import lib as L, synth
synth.DRIFT_TARGET = 0.05
g = synth.manhattan()
def mk():
    G = L.Graph(fix_small_angle_b=1, pcg_rel_tol=1e-8)
    G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"]); G.initialize()
    G.optimize(2); G.set_vertices(g["states"])
    return G
Gs = [mk() for _ in range(4)]
for T in (1, 2, 3, 4, 1, 2):
    for G in Gs[:T]: G.set_vertices(g["states"])
    th = [threading.Thread(target=lambda G=G: G.optimize(8)) for G in Gs[:T]]
    t = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    dt = time.perf_counter() - t
    print("T = %d concurrent runs of 8 LM iterations: %.1f ms = %.2f x one run; per run %.1f ms; pcg %s" % (
        T, dt * 1e3, 0, dt * 1e3 / T, [s.pcg_iters for s in Gs[0].stats()][:3]), flush=True)
