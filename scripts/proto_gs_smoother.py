"""CPU experiment (numpy / scipy; scripts/proto_amg.py's system and hierarchy): would a block Gauss-Seidel smoother on
level 0 -- forward sweep from zero before the coarse correction, backward sweep after it: the symmetric pair PCG needs --
cut the PCG iterations of the product's cycle (damped block-Jacobi, omega 0.9, over-correction 1.8 / 1.6, cycle 2/3/3)
by enough to pay for what it costs on a GPU (a sweep is sequential between colours: one launch per colour)?
Orderings: natural (the best case for Gauss-Seidel, not parallel) and multicolour (greedy colouring, rows grouped by
colour: what a GPU would run).  Prints PCG iterations per variant.  python scripts/proto_gs_smoother.py [V=6000]"""
import os, sys, time
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proto_amg as PA
from sim3opt_amd import synth, sim3np as S3

V = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
side = int(round((V / 10) ** 0.5))
g = synth.manhattan(V, 10 * V, dims=(side, side, 10))
rng = np.random.default_rng(0)


def colouring(adj):
    n = adj.shape[0]
    col = -np.ones(n, dtype=np.int64)
    indptr, indices = adj.indptr, adj.indices
    for i in range(n):
        used = set(col[indices[indptr[i]:indptr[i + 1]]].tolist())
        c = 0
        while c in used:
            c += 1
        col[i] = c
    return col


def cycle(levels, k, r, visits, over, smooth0):
    """The product's cycle: levels[k] solved by one visit; W-type revisits of level k + 1 (visits[k + 1]) with a
    pre-smoothing step between them.  smooth0: (pre, post) callables for level 0, else damped block-Jacobi."""
    L = levels[k]
    if k == len(levels) - 1:
        return L.lu.solve(r)
    if k == 0 and smooth0 is not None:
        x = smooth0[0](r)
    else:
        x = L.omega * PA.bj(L, r)
    rc = L.P.T @ (r - L.A @ x)
    Lc = levels[k + 1]
    xc = cycle(levels, k + 1, rc, visits, over, smooth0)
    for _ in range(visits[k + 1] - 1):
        if k + 1 < len(levels) - 1:
            xc = xc + Lc.omega * PA.bj(Lc, rc - Lc.A @ xc)        # pre-smoothing step of the revisit
            xc = xc + cycle_correction(levels, k + 1, rc - Lc.A @ xc, visits, over)
    x = x + over[0 if k == 0 else 1] * (L.P @ xc)
    if k == 0 and smooth0 is not None:
        return smooth0[1](x, r)
    return x + L.omega * PA.bj(L, r - L.A @ x)


def cycle_correction(levels, k, r, visits, over):
    """coarse correction + post-smoothing of a revisit (no pre-smoothing of its own: done by the caller)"""
    L = levels[k]
    rc = L.P.T @ r
    xc = cycle(levels, k + 1, rc, visits, over, None) if k + 1 < len(levels) else None
    x = over[1] * (L.P @ xc)
    return x + L.omega * PA.bj(L, r - L.A @ x)


for label, states, lam_rel in (("initial state, lambda = 1e-5 max diag", g["states"], 1e-5), ("near the optimum, lambda = 1e-8 max diag", None, 1e-8)):
    if states is None:
        xi = rng.standard_normal((V, 7)) * np.array([1e-3] * 3 + [1e-2] * 3 + [1e-3])
        states = S3.mul(S3.exp(xi, fix_b=True), g["gt"])
    H, rhs, adj, free = PA.build_system(g, states)
    lam = lam_rel * H.diagonal().max()
    A = (H + lam * sp.identity(H.shape[0])).tocsr()
    lv = PA.build_hierarchy(H, lam, adj, states[free], 3, 3, 200, 0.9, np.random.default_rng(0))  # (three levels: level 2 exact)
    print("== %s: levels %s" % (label, [l.A.shape[0] // 7 for l in lv]), flush=True)
    visits = [1, 2, 3, 3, 3]
    over = (1.8, 1.6)
    nb = A.shape[0] // 7
    blk = sp.kron(sp.csr_matrix(adj + sp.identity(nb)), np.ones((7, 7))).tocsr()   # block pattern

    def gs_pair(order):
        """forward / backward block Gauss-Seidel in the given row order (a permutation of the block rows)"""
        perm = (7 * np.repeat(order, 7).reshape(-1, 7) + np.arange(7)).ravel()
        Ap = A[perm][:, perm].tocsr()
        pos = np.empty(nb, dtype=np.int64); pos[np.arange(nb)] = np.arange(nb)
        rb = np.repeat(np.arange(nb), 7)
        coo = Ap.tocoo()
        low = coo.row // 7 >= coo.col // 7
        Lo = sp.csc_matrix((coo.data[low], (coo.row[low], coo.col[low])), shape=Ap.shape)       # D + L (block lower)
        Up = sp.csc_matrix((coo.data[~low | (coo.row // 7 == coo.col // 7)], (coo.row[~low | (coo.row // 7 == coo.col // 7)], coo.col[~low | (coo.row // 7 == coo.col // 7)])), shape=Ap.shape)  # D + U
        luL = spla.splu(Lo, permc_spec="NATURAL", diag_pivot_thresh=0.0)
        luU = spla.splu(Up, permc_spec="NATURAL", diag_pivot_thresh=0.0)
        inv = np.empty_like(perm); inv[perm] = np.arange(perm.size)

        def pre(r):
            return luL.solve(r[perm])[inv]

        def post(x, r):
            xp, rp = x[perm], r[perm]
            return (xp + luU.solve(rp - Ap @ xp))[inv]
        return pre, post

    def run(name, smooth0):
        t = time.time()
        _, it = PA.pcg(A, rhs, lambda r: cycle(lv, 0, r, visits, over, smooth0), 1e-8, 400)
        print("  %-58s %3d PCG iterations (%.1f s)" % (name, it, time.time() - t), flush=True)

    run("damped block-Jacobi on every level (the product)", None)
    run("level 0: block Gauss-Seidel, natural order", gs_pair(np.arange(nb)))
    col = colouring(sp.csr_matrix(adj))
    order = np.argsort(col, kind="stable")
    print("  multicolour: %d colours, sizes %s" % (col.max() + 1, np.bincount(col).tolist()))
    run("level 0: block Gauss-Seidel, multicolour order", gs_pair(order))
