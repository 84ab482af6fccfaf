"""Profiling target: 50 exact solves on each KITTI-00 graph (run under rocprofv3 --kernel-trace --stats)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sim3opt_amd import lib as L
import kitti_graph as K
which = sys.argv[1] if len(sys.argv) > 1 else "both"
for name, one in (("one", True), ("all", False)):
    if which not in ("both", name): continue
    g = K.build_direct_graph(one)
    G = L.Graph(verbose=1)
    G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"]); G.initialize()
    G.linearize()
    for _ in range(5): G.solve(1.0)
    t = time.perf_counter()
    for _ in range(50): G.solve(1.0)
    print(name, "solve: %.1f us" % (1e6 * (time.perf_counter() - t) / 50), flush=True)
