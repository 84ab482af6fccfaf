"""Per-trial lambda / PCG iterations of LM iterations 1..12 on config 3 (verbose = 2)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sim3opt_amd import lib as L, synth
synth.DRIFT_TARGET = 0.05
g = synth.manhattan()
G = L.Graph(fix_small_angle_b=1, pcg_rel_tol=1e-8, verbose=2)
G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"]); G.initialize()
G.optimize(12)
