"""Profiling target for the batched solve: config 3, 11 LM iterations from the initial state (LM iteration 10 is
the burst of eight trials: one 4-system batch of 41 iterations).  Run under rocprofv3 (--pmc ...)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sim3opt_amd import lib as L, synth
synth.DRIFT_TARGET = 0.05
g = synth.manhattan(100000, 1000000)
G = L.Graph(fix_small_angle_b=1, pcg_rel_tol=1e-8)
G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"]); G.initialize()
n = G.optimize(11)
kt = G.kernel_times()
print("iterations", n, "trials", [s.trials for s in G.stats()], "batches", kt.n_batches, "systems", kt.n_batched_solves)
