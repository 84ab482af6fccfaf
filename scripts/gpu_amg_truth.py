"""True (2-norm) residual at the end of multigrid-preconditioned solves, healthy and over-corrected."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sim3opt_amd import lib as L, synth
synth.DRIFT_TARGET = 0.05
g = synth.manhattan(100000, 1000000)
for cfg in sys.argv[1:] or [""]:
    for k in ("SIM3OPT_AMG_CYCLE", "SIM3OPT_AMG_OVER"): os.environ.pop(k, None)
    for kv in cfg.split():
        k, v = kv.split("="); os.environ["SIM3OPT_AMG_" + k] = v
    print("==== config:", cfg or "(default)", flush=True)
    G = L.Graph(fix_small_angle_b=1, pcg_rel_tol=1e-8, verbose=1)
    G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"]); G.initialize()
    G.optimize(8)
    G.close()
