import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, scipy.sparse as sp
from sim3opt_amd import lib as L, synth
synth.DRIFT_TARGET = 0.05
g = synth.chain_loop(2000, 4000)
for pre in (0, 1):
    G = L.Graph(fix_small_angle_b=1, fd_delta=1e-6, pcg_rel_tol=1e-10, pcg_max_iters=20000, preconditioner=pre)
    G.add_vertices(g['states'], g['fixed']); G.add_edges(g['v0'], g['v1'], g['meas']); G.initialize(); G.linearize()
    rowptr, colidx, blocks, b = G.get_system()
    M = sp.bsr_matrix((blocks, colidx, rowptr), blocksize=(7, 7)).tocsr()
    for lamf in (1e-5, 1e-9):
        lam = lamf * abs(M.diagonal()).max()
        x, it, rr = G.solve(lam)
        res = np.linalg.norm(b - (M @ x + lam * x)) / np.linalg.norm(b)
        print("pre", pre, "lam %.0e" % lamf, "iters", it, "relres %.1e" % rr, "true res %.1e" % res)
