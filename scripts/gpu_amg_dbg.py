import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sim3opt_amd import lib as L, synth
V, E, dims = int(sys.argv[1]), int(sys.argv[2]), (int(sys.argv[3]), int(sys.argv[3]), 10)
g = synth.manhattan(V, E, dims=dims)
G = L.Graph(fix_small_angle_b=1, pcg_rel_tol=1e-8, preconditioner=2, verbose=1)
G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"]); G.initialize()
t = time.perf_counter(); G.optimize(int(sys.argv[4]) if len(sys.argv) > 4 else 3); print("s", time.perf_counter() - t)
