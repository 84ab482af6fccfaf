"""Tuning aid: per-level time stamps of the top group of the exact factorisation (SIM3OPT_DIRECT_TRACE)."""
import os, sys
os.environ["SIM3OPT_DIRECT_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sim3opt_amd import lib as L
import kitti_graph as K
for one in (True, False):
    g = K.build_direct_graph(one)
    G = L.Graph()
    G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"]); G.initialize()
    G.linearize()
    for _ in range(3): G.solve(1.0)
