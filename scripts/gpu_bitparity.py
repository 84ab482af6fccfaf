"""How close is the device evaluation of residuals / numeric Jacobians to the CPU oracle's, bit for bit?
KITTI-00 at the oracle's states after 0..4 LM iterations (reference configuration)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle as O
from sim3opt_amd import lib as L
import kitti_graph as K
for one in (True, False):
    g = K.build_direct_graph(one)
    OG = O.Graph(g["states"], g["fixed"], g["v0"], g["v1"], g["meas"])
    G = L.Graph(); G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"]); G.initialize()
    lam = 0.0
    for k in range(6):
        G.set_vertices(OG.states)
        e_g, e_c = G.edge_errors(), OG.errors()
        G.linearize()
        Hg, bg = G.dense_system()
        Hc, bc = OG.build_dense()
        nz = Hc != 0
        print("one" if one else "all", "state", k, "e: bit-equal %.4f max|d| %.2e | H: bit-equal %.4f max|dH|/max|H| %.2e max|H| %.2e | b rel %.2e" % (
            (e_g == e_c).mean(), np.abs(e_g - e_c).max(), (Hg == Hc)[nz].mean(), np.abs(Hg - Hc).max() / np.abs(Hc).max(),
            np.abs(Hc).max(), np.abs(bg - bc).max() / np.abs(bc).max()), flush=True)
        it, tr = OG.optimize(1, O.default_options(user_lambda_init=lam))
        lam = tr[0].lambda_
