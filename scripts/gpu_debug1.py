import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oracle import oracle as O
from sim3opt_amd import lib as L, sim3np as S3
import kitti_graph as K

for one in (True, False):
    g = K.build_direct_graph(one)
    OG = O.Graph(g['states'], g['fixed'], g['v0'], g['v1'], g['meas'])
    G = L.Graph()
    G.add_vertices(g['states'], g['fixed'])
    G.add_edges(g['v0'], g['v1'], g['meas'])
    G.initialize()
    e_gpu = G.edge_errors(); e_or = OG.errors()
    print("one", one, "edge err max diff", np.abs(e_gpu - e_or).max(), "chi2 gpu", G.chi2(), "oracle", OG.chi2())
    G.linearize()
    H, b = G.dense_system()
    Ho, bo = OG.build_dense()
    print(" H diff", np.abs(H - Ho).max(), "rel", np.abs(H - Ho).max() / np.abs(Ho).max(), "b diff", np.abs(b - bo).max(), "sym", np.abs(H - H.T).max())
    lam = 1e-5 * np.abs(np.diag(Ho)).max()
    t = time.time(); x, it, rr = G.solve(lam); dt = time.time() - t
    ok, xo, _ = OG.solve_once(lam)
    xd = np.linalg.solve(H + lam * np.eye(H.shape[0]), b)
    print(" solve iters", it, "relres", rr, "time", dt, "x vs oracle", np.abs(x - xo).max() / np.abs(xo).max(), "x vs dense(H_gpu)", np.abs(x - xd).max() / np.abs(xd).max())
    G.set_options(verbose=0)
    t = time.time(); n = G.optimize(20); dt = time.time() - t
    st = G.stats()
    print(" gpu iters", n, "time", dt)
    print(" gpu chi2:", [round(s.chi2_after, 6) for s in st])
    print(" gpu trials:", [s.trials for s in st], "pcg", [s.pcg_iters for s in st])
    ito, tr = OG.optimize(20)
    print(" ora chi2:", [round(s.chi2_after, 6) for s in tr])
    print(" ora trials:", [s.trials for s in tr])
    pg = S3.inv(G.get_vertices())[:, 4:7]; po = S3.inv(OG.states)[:, 4:7]
    print(" rmse gpu vs oracle after 20 it", np.sqrt(((pg - po) ** 2).sum(1).mean()))
    print(" ms lin/solve/upd", [(round(s.ms_linearize, 3), round(s.ms_solve, 3), round(s.ms_update, 3)) for s in st[:3]])
