"""Profiling target: KITTI-00 in the reference's configuration (kitti_surf.cpp:674-675), optimize(100) on the
one-loop and on the 118-loop graph (run under rocprofv3 --kernel-trace --stats for the per-kernel split).
Prints wall time per LM iteration and per trial."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sim3opt_amd import lib as L
import kitti_graph as K
which = sys.argv[1] if len(sys.argv) > 1 else "one"
for name, one in (("one", True), ("all", False)):
    if which not in ("both", name): continue
    g = K.build_direct_graph(one)
    G = L.Graph()
    G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"]); G.initialize()
    G.optimize(100)                      # warm-up (caches, first launches)
    G.set_vertices(g["states"])
    t = time.perf_counter()
    n = G.optimize(100)
    dt = time.perf_counter() - t
    tr = sum(s.trials for s in G.stats())
    print("%s: %d LM iterations, %d trials, %.2f ms: %.1f us per iteration, %.1f us per trial" %
          (name, n, tr, 1e3 * dt, 1e6 * dt / n, 1e6 * dt / tr), flush=True)
