"""One-off parity check at a size where the hierarchy has three levels and the oracle's exact LDL^T
is still affordable: Manhattan 3000 vertices / 30000 edges, 4 LM iterations, delta = 1e-6."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as O
from sim3opt_amd import lib as L, synth
synth.DRIFT_TARGET = 0.05
g = synth.manhattan(3000, 30000, dims=(17, 17, 10))
G = L.Graph(fix_small_angle_b=1, fd_delta=1e-6, pcg_rel_tol=1e-12)
G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"]); G.initialize()
print("levels", G.amg_hierarchy()[0], "preconditioner", G.preconditioner_in_use(), flush=True)
t = time.perf_counter(); G.optimize(4); print("gpu %.3fs" % (time.perf_counter() - t), [s.pcg_iters for s in G.stats()], flush=True)
OG = O.Graph(g["states"], g["fixed"], g["v0"], g["v1"], g["meas"])
t = time.perf_counter(); it, tr = OG.optimize(4, O.default_options(fix_small_angle_b=1, fd_delta=1e-6, threads=8)); print("oracle %.1fs" % (time.perf_counter() - t), flush=True)
print("chi2 gpu", [s.chi2_after for s in G.stats()])
print("chi2 cpu", [t_.chi2_after for t_ in tr])
print("trajectory RMSE gpu vs oracle %.3e" % synth.rmse(G.get_vertices(), OG.states))
