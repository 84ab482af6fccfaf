"""Profiling target: optimize(100) on the KITTI-00 one-loop graph in the reference configuration
(run under rocprofv3 --kernel-trace; scripts/trace_gaps.py reads the trace)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sim3opt_amd import lib as L
import kitti_graph as K
g = K.build_direct_graph(True)
G = L.Graph()
G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"]); G.initialize()
G.optimize(2); G.set_vertices(g["states"])
t = time.perf_counter(); n = G.optimize(100)
print("iterations", n, "seconds %.4f" % (time.perf_counter() - t), "trials", sum(s.trials for s in G.stats()))
