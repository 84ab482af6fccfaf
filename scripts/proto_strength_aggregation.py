"""CPU experiment: aggregation by numerical strength of connection (||D_i^-1/2 A_ij D_j^-1/2||_F) instead of by block counts."""
import os, sys, time
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
sys.path.insert(0, "/root/repo/scripts"); sys.path.insert(0, "/root/repo")
import proto_amg as PA
from sim3opt_amd import synth, sim3np as S3
src = open("/root/repo/scripts/proto_gs_smoother.py").read()
ns = {"PA": PA, "np": np}
exec(src[src.index("def cycle("):src.index("for label, states, lam_rel")], ns)
cycle = ns["cycle"]
V = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
side = int(round((V / 10) ** 0.5))
g = synth.manhattan(V, 10 * V, dims=(side, side, 10))
rng = np.random.default_rng(0)
for label, states, lam_rel in (("initial", g["states"], 1e-5), ("near optimum", None, 1e-8)):
    if states is None:
        xi = rng.standard_normal((V, 7)) * np.array([1e-3] * 3 + [1e-2] * 3 + [1e-3])
        states = S3.mul(S3.exp(xi, fix_b=True), g["gt"])
    H, rhs, adj, free = PA.build_system(g, states)
    lam = lam_rel * H.diagonal().max()
    A = (H + lam * sp.identity(H.shape[0])).tocsr()
    Hb = sp.bsr_matrix(A, blocksize=(7, 7))
    nb = Hb.shape[0] // 7
    # strength: Frobenius norm of the block scaled by the diagonal blocks' traces
    dtr = np.zeros(nb)
    rows = np.repeat(np.arange(nb), np.diff(Hb.indptr))
    isd = rows == Hb.indices
    dtr[rows[isd]] = np.trace(Hb.data[isd], axis1=1, axis2=2)
    w = np.linalg.norm(Hb.data.reshape(-1, 49), axis=1) / np.sqrt(dtr[rows] * dtr[Hb.indices])
    strength = sp.csr_matrix((np.where(isd, 0.0, w), Hb.indices, Hb.indptr), shape=(nb, nb))
    for name, graph in (("block counts (the product)", adj), ("numerical strength", strength)):
        lv = PA.build_hierarchy(H, lam, graph, states[free], 3, 3, 200, 0.9, np.random.default_rng(0))
        _, it = PA.pcg(A, rhs, lambda r: cycle(lv, 0, r, [1, 2, 3, 3, 3], (1.8, 1.6), None), 1e-8, 400)
        print(f"{label}: aggregation by {name}: levels {[l.A.shape[0] // 7 for l in lv]} blocks {[l.nnzb for l in lv]}: {it} PCG iterations", flush=True)
