import os, sys
sys.path.insert(0, os.getcwd())
from sim3opt_amd import lib as L, synth
synth.DRIFT_TARGET = 0.05
g = synth.manhattan(100000, 1000000)
G = L.Graph(fix_small_angle_b=1, pcg_rel_tol=1e-8, verbose=int(os.environ.get("VERB", "0")))
G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"]); G.initialize()
G.optimize(5); G.set_vertices(g["states"])
n = G.optimize(20)
st = G.stats()
print("trials", [s.trials for s in st], "pcg", [s.pcg_iters for s in st])
print("ms_solve", [round(s.ms_solve, 2) for s in st], "total solve ms", round(sum(s.ms_solve for s in st), 1), "chi2", st[-1].chi2_after)
