"""CPU experiment: prolongation smoothed INSIDE the tentative pattern (P_i <- P_i - w D_i^-1 sum_{j in the same aggregate} A_ij P_j): level 1
keeps its size.  Does it cut the PCG iterations?"""
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
    for w in (0.0, 0.3, 0.5, 0.67, 1.0):
        lv = PA.build_hierarchy(H, lam, adj, states[free], 3, 3, 200, 0.9, np.random.default_rng(0))
        if w > 0:
            L0 = lv[0]
            Dinv = sp.block_diag([L0.Dinv[i] for i in range(L0.Dinv.shape[0])], format="csr")
            P0 = L0.P.tocsr()
            mask = (P0 != 0).astype(float)
            # block pattern mask: keep whole 7x7 blocks of the tentative pattern
            nb, nc = P0.shape[0] // 7, P0.shape[1] // 7
            aggr = np.asarray(sp.bsr_matrix(P0, blocksize=(7, 7)).indices)
            Mb = sp.kron(sp.csr_matrix((np.ones(nb), (np.arange(nb), aggr)), shape=(nb, nc)), np.ones((7, 7))).tocsr()
            Pn = P0 - w * (Dinv @ ((A - sp.block_diag([np.linalg.inv(L0.Dinv[i]) for i in range(nb)], format="csr")) @ P0)).multiply(Mb)
            L0.P = sp.csr_matrix(Pn)
            A1 = (L0.P.T @ A @ L0.P).tobsr(blocksize=(7, 7))
            lv[1].A = A1; lv[1].Dinv = PA.block_diag_inv(A1, 0.0); lv[1].nnzb = A1.nnz // 49
            A2 = (lv[1].P.T @ sp.csr_matrix(A1) @ lv[1].P).tobsr(blocksize=(7, 7))
            lv[2].A = A2; lv[2].lu = spla.splu(sp.csc_matrix(A2)); lv[2].nnzb = A2.nnz // 49
        for over in ((1.8, 1.6), (1.4, 1.6), (1.0, 1.6)):
            _, it = PA.pcg(A, rhs, lambda r: cycle(lv, 0, r, [1, 2, 3, 3, 3], over, None), 1e-8, 400)
            print(f"{label}: in-pattern smoothing w = {w}: level blocks {[l.nnzb for l in lv]}, over-correction {over}: {it} PCG iterations", flush=True)
