"""CPU experiment (numpy / scipy, scripts/proto_amg.py + the cycle of scripts/proto_gs_smoother.py): smoothed aggregation on level 0,
P = (I - w D^-1 A) P_tentative, against the product's piecewise-constant (adjoint-transported) prolongation: PCG iterations and the size
of what the cycle would have to stream.  python scripts/proto_smoothed_aggregation.py [V=4000]"""
import os, sys, time
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
sys.path.insert(0, "/root/repo/scripts"); sys.path.insert(0, "/root/repo")
import proto_amg as PA
import importlib.util
from sim3opt_amd import synth, sim3np as S3
V = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
side = int(round((V / 10) ** 0.5))
g = synth.manhattan(V, 10 * V, dims=(side, side, 10))
rng = np.random.default_rng(0)
spec = importlib.util.spec_from_file_location("gs", "/root/repo/scripts/proto_gs_smoother.py")
# reuse cycle() from the GS script without running its main: copy the functions
src = open("/root/repo/scripts/proto_gs_smoother.py").read()
ns = {"PA": PA, "np": np}
exec(src[src.index("def cycle("):src.index("for label, states, lam_rel")], ns)
cycle = ns["cycle"]; 
import types
# cycle_correction references cycle via globals of exec namespace
for label, states, lam_rel in (("initial", g["states"], 1e-5), ("near optimum", None, 1e-8)):
    if states is None:
        xi = rng.standard_normal((V, 7)) * np.array([1e-3] * 3 + [1e-2] * 3 + [1e-3])
        states = S3.mul(S3.exp(xi, fix_b=True), g["gt"])
    H, rhs, adj, free = PA.build_system(g, states)
    lam = lam_rel * H.diagonal().max()
    A = (H + lam * sp.identity(H.shape[0])).tocsr()
    for sa_omega in (0.0, 0.5, 0.67):
        lv = PA.build_hierarchy(H, lam, adj, states[free], 3, 3, 200, 0.9, np.random.default_rng(0))
        if sa_omega > 0:
            L0 = lv[0]
            Dinv = sp.block_diag([L0.Dinv[i] for i in range(L0.Dinv.shape[0])], format="csr")
            P = (sp.identity(A.shape[0]) - sa_omega * Dinv @ A) @ L0.P
            L0.P = P.tocsr()
            A1 = (L0.P.T @ A @ L0.P).tobsr(blocksize=(7, 7))
            lv[1].A = A1; lv[1].Dinv = PA.block_diag_inv(A1, 0.0); lv[1].nnzb = A1.nnz // 49
            A2 = (lv[1].P.T @ sp.csr_matrix(A1) @ lv[1].P).tobsr(blocksize=(7, 7))
            lv[2].A = A2; lv[2].lu = spla.splu(sp.csc_matrix(A2)); lv[2].nnzb = A2.nnz // 49
        for over in ((1.8, 1.6), (1.0, 1.6), (1.0, 1.0)):
            _, it = PA.pcg(A, rhs, lambda r: cycle(lv, 0, r, [1, 2, 3, 3, 3], over, None), 1e-8, 400)
            print(f"{label}: SA omega {sa_omega}: level blocks {[l.nnzb for l in lv]}, P blocks {lv[0].P.nnz // 49}, over {over}: {it} PCG iterations", flush=True)
