import os, sys, time
sys.path.insert(0, "/root/repo")
from sim3opt_amd import lib as L, synth
synth.DRIFT_TARGET = 0.05
g = synth.manhattan(100000, 1000000)
G = L.Graph(fix_small_angle_b=1, pcg_rel_tol=1e-8, verbose=2)
G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"]); G.initialize()
G.optimize(11)
