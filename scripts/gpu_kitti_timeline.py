"""Profiling target: optimize(N) on the one-loop KITTI-00 graph in the reference's configuration (run under
rocprofv3 --kernel-trace; scripts/kitti_timeline_report.py prints the launch sequence of one LM iteration)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sim3opt_amd import lib as L
import kitti_graph as K
one = (sys.argv[1] if len(sys.argv) > 1 else "one") == "one"
g = K.build_direct_graph(one)
G = L.Graph()
G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"]); G.initialize()
G.optimize(5); G.set_vertices(g["states"])
t = time.perf_counter(); n = G.optimize(40); dt = time.perf_counter() - t
st = G.stats()
print("%d iterations, %.1f us each, trials %s" % (n, 1e6 * dt / n, [s.trials for s in st][:12]), flush=True)
