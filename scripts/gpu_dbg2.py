import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle as O
from sim3opt_amd import lib as L
import kitti_graph as K
g = K.build_direct_graph(True)
OG = O.Graph(g['states'], g['fixed'], g['v0'], g['v1'], g['meas']); it, tr = OG.optimize(3)
print("oracle", [t.chi2_after for t in tr], [t.trials for t in tr])
for pre in (0, 1):
    for tol in (1e-12, 1e-14):
        G = L.Graph(pcg_rel_tol=tol, pcg_max_iters=60000, preconditioner=pre)
        G.add_vertices(g['states'], g['fixed']); G.add_edges(g['v0'], g['v1'], g['meas']); G.initialize()
        G.optimize(3); st = G.stats()
        print("LM pre", pre, "tol", tol, [s.chi2_after for s in st], [s.trials for s in st], [s.pcg_iters for s in st], ["%.1e" % s.pcg_rel_res for s in st])
# linear-solve accuracy at iteration 2's system
G = L.Graph(pcg_rel_tol=1e-12, pcg_max_iters=60000, preconditioner=0)
G.add_vertices(g['states'], g['fixed']); G.add_edges(g['v0'], g['v1'], g['meas']); G.initialize(); G.optimize(1)
lam = G.stats()[0].lambda_
G.linearize(); H, b = G.dense_system()
xd = np.linalg.solve(H + lam * np.eye(len(b)), b)
print("cond", np.linalg.cond(H + lam * np.eye(len(b))), "lam", lam)
for pre in (0, 1):
    G2 = L.Graph(pcg_rel_tol=1e-12, pcg_max_iters=60000, preconditioner=pre)
    G2.add_vertices(G.get_vertices(), g['fixed']); G2.add_edges(g['v0'], g['v1'], g['meas']); G2.initialize(); G2.linearize()
    x, it, rr = G2.solve(lam)
    print("pre", pre, "iters", it, "rr %.1e" % rr, "x err vs dense %.2e" % (np.abs(x - xd).max() / np.abs(xd).max()), "true res %.1e" % (np.linalg.norm(b - H @ x - lam * x) / np.linalg.norm(b)))
