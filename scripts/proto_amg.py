"""CPU prototype (numpy/scipy, uses the oracle's Jacobians -- test infrastructure, not product):
does an aggregation multigrid V-cycle with adjoint-transported prolongation beat block-Jacobi as the
PCG preconditioner on Manhattan pose graphs?  Prints PCG iteration counts and the work per iteration
in fine-SpMV equivalents.  Run here (no GPU): python scripts/proto_amg.py [V] [E]
"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
from sim3opt_amd import synth, sim3np as S3  # noqa: E402
import oracle as O  # noqa: E402


def adjoint(S):
    """Ad_S (n,7,7), tangent order [omega, upsilon, sigma]:  S exp(x) S^-1 = exp(Ad_S x)."""
    R = S3.quat_to_R(S[:, :4])
    t = S[:, 4:7]
    s = S[:, 7]
    n = S.shape[0]
    Ad = np.zeros((n, 7, 7))
    tx = np.zeros((n, 3, 3))
    tx[:, 0, 1], tx[:, 0, 2] = -t[:, 2], t[:, 1]
    tx[:, 1, 0], tx[:, 1, 2] = t[:, 2], -t[:, 0]
    tx[:, 2, 0], tx[:, 2, 1] = -t[:, 1], t[:, 0]
    Ad[:, :3, :3] = R
    Ad[:, 3:6, :3] = tx @ R
    Ad[:, 3:6, 3:6] = s[:, None, None] * R
    Ad[:, 3:6, 6] = -t
    Ad[:, 6, 6] = 1.0
    return Ad


def build_system(g, states, fix_b=1):
    G = O.Graph(states, g["fixed"], g["v0"], g["v1"], g["meas"])
    opt = O.default_options(fix_small_angle_b=fix_b)
    A, B = G.jacobians(opt)
    e = G.errors(opt)
    V = states.shape[0]
    free = np.where(g["fixed"] == 0)[0]
    hidx = -np.ones(V, dtype=np.int64)
    hidx[free] = np.arange(free.size)
    a, b = hidx[g["v0"]], hidx[g["v1"]]
    nb = free.size
    rows, cols, blocks = [], [], []
    for (i, Ji), (j, Jj) in (((a, A), (a, A)), ((b, B), (b, B)), ((a, A), (b, B)), ((b, B), (a, A))):
        m = (i >= 0) & (j >= 0)
        rows.append(i[m]); cols.append(j[m])
        blocks.append(np.einsum("kri,krj->kij", Ji[m], Jj[m]))
    rows = np.concatenate(rows); cols = np.concatenate(cols); blocks = np.concatenate(blocks)
    # expand to scalar COO
    r = (7 * rows[:, None, None] + np.arange(7)[None, :, None]).repeat(7, axis=2)
    c = (7 * cols[:, None, None] + np.arange(7)[None, None, :]).repeat(7, axis=1)
    H = sp.coo_matrix((blocks.ravel(), (r.ravel(), c.ravel())), shape=(7 * nb, 7 * nb)).tocsr()
    H = sp.bsr_matrix(H, blocksize=(7, 7))
    rhs = np.zeros(7 * nb)
    for i, J in ((a, A), (b, B)):
        m = i >= 0
        np.add.at(rhs.reshape(nb, 7), i[m], -np.einsum("kri,kr->ki", J[m], e[m]))
    adj = sp.coo_matrix((np.ones(rows.size), (rows, cols)), shape=(nb, nb)).tocsr()
    return H, rhs, adj, free


def pairwise_aggregate(adj, passes, rng):
    """`passes` rounds of greedy matching on the (coarsened) graph; returns fine->aggregate map."""
    n = adj.shape[0]
    agg = np.arange(n)
    A = adj.copy()
    for _ in range(passes):
        A = A.tocsr()
        A.setdiag(0)
        A.eliminate_zeros()
        m = A.shape[0]
        match = -np.ones(m, dtype=np.int64)
        indptr, indices, data = A.indptr, A.indices, A.data
        for i in range(m):  # natural order: deterministic
            if match[i] >= 0:
                continue
            best, bw = -1, 0.0
            for k in range(indptr[i], indptr[i + 1]):
                j = indices[k]
                if match[j] < 0 and j != i and data[k] > bw:
                    best, bw = j, data[k]
            if best >= 0:
                match[i] = best
                match[best] = i
            else:
                match[i] = i
        rep = np.minimum(np.arange(m), match)
        uniq, cid = np.unique(rep, return_inverse=True)
        agg = cid[agg]
        P = sp.coo_matrix((np.ones(m), (np.arange(m), cid)), shape=(m, uniq.size)).tocsr()
        A = (P.T @ A @ P).tocsr()
    return agg, int(agg.max()) + 1


def block_diag_inv(H, lam):
    nb = H.shape[0] // 7
    D = np.zeros((nb, 7, 7))
    Hb = H.tobsr(blocksize=(7, 7))
    for i in range(nb):
        for k in range(Hb.indptr[i], Hb.indptr[i + 1]):
            if Hb.indices[k] == i:
                D[i] += Hb.data[k]
    D += lam * np.eye(7)
    return np.linalg.inv(D)


class Level:
    pass


def build_hierarchy(H, lam, adj, states_free, agg_passes, max_levels, coarse_size, omega, rng):
    levels = []
    A = (H + lam * sp.identity(H.shape[0])).tobsr(blocksize=(7, 7))
    first = True
    while True:
        L = Level()
        L.A = A
        L.Dinv = block_diag_inv(A, 0.0)
        L.nnzb = A.nnz // 49
        levels.append(L)
        nb = A.shape[0] // 7
        if nb <= coarse_size or len(levels) >= max_levels:
            L.lu = spla.splu(sp.csc_matrix(A))
            break
        agg, nc = pairwise_aggregate(adj, agg_passes, rng)
        # prolongation: first level transports the coarse (world-frame) variable by Ad_S of the
        # fine vertex; deeper levels are piecewise constant
        if first:
            Pb = adjoint(states_free)
            first = False
        else:
            Pb = np.tile(np.eye(7), (nb, 1, 1))
        P = sp.bsr_matrix((Pb, agg, np.arange(nb + 1)), shape=(7 * nb, 7 * nc))
        L.P = P.tocsr()
        A = (L.P.T @ sp.csr_matrix(A) @ L.P).tobsr(blocksize=(7, 7))
        Pa = sp.coo_matrix((np.ones(nb), (np.arange(nb), agg)), shape=(nb, nc)).tocsr()
        adj = (Pa.T @ adj @ Pa).tocsr()
    for L in levels:
        L.omega = omega
    return levels


def bj(L, r):
    return np.einsum("kij,kj->ki", L.Dinv, r.reshape(-1, 7)).ravel()


ADDITIVE0 = False


def vcycle(levels, k, r, nu=1, gamma=1, over=1.0):
    L = levels[k]
    if k == len(levels) - 1:
        return L.lu.solve(r)
    if k == 0 and ADDITIVE0:  # no fine-level matrix pass: M^-1 = omega D^-1 + P V_1 P^T
        rc = L.P.T @ r
        xc = vcycle(levels, 1, rc, nu, gamma, over)
        if gamma == 2 and 1 < len(levels) - 1:
            xc += vcycle(levels, 1, rc - levels[1].A @ xc, nu, gamma, over)
        return bj(L, r) + over * (L.P @ xc)
    x = L.omega * bj(L, r)
    for _ in range(nu - 1):
        x += L.omega * bj(L, r - L.A @ x)
    rc = L.P.T @ (r - L.A @ x)
    xc = vcycle(levels, k + 1, rc, nu, gamma, over)
    if gamma == 2 and k + 1 < len(levels) - 1:
        Lc = levels[k + 1]
        xc += vcycle(levels, k + 1, rc - Lc.A @ xc, nu, gamma, over)
    x += over * (L.P @ xc)
    for _ in range(nu):
        x += L.omega * bj(L, r - L.A @ x)
    return x


def pcg(A, b, M, tol, maxit):
    x = np.zeros_like(b)
    r = b.copy()
    z = M(r)
    p = z.copy()
    rz = r @ z
    rz0 = rz
    for it in range(1, maxit + 1):
        q = A @ p
        alpha = rz / (p @ q)
        x += alpha * p
        r -= alpha * q
        z = M(r)
        rzn = r @ z
        if rzn <= tol * tol * rz0:
            return x, it
        p = z + (rzn / rz) * p
        rz = rzn
    return x, maxit


CONFIGS = ((3, 0.8, 1, 1, 1.0), (3, 0.8, 1, 2, 1.0), (-3, 0.8, 1, 1, 1.0), (-3, 0.8, 1, 2, 1.0), (-2, 0.8, 1, 2, 1.0))


def main():
    global ADDITIVE0
    kind = sys.argv[1] if len(sys.argv) > 1 else "20000"
    if kind.startswith("kitti"):
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
        import kitti_graph as K
        g = K.build_direct_graph(kind == "kitti1")
        g["gt"] = g["states"]
        V = g["states"].shape[0]
    elif kind == "chain":
        g = synth.chain_loop()
        V = g["states"].shape[0]
    else:
        V = int(kind)
        E = int(sys.argv[2]) if len(sys.argv) > 2 else 10 * V
        side = int(round((V / 10) ** 0.5))
        g = synth.manhattan(V, E, dims=(side, side, 10))
    rng = np.random.default_rng(0)
    for label, states, lam_rel in (("initial, lam=1e-5 max", g["states"], 1e-5),
                                   ("near gt, lam=1e-8 max", None, 1e-8)):
        if states is None:
            xi = rng.standard_normal((V, 7)) * np.array([1e-3] * 3 + [1e-2] * 3 + [1e-3])
            states = S3.mul(S3.exp(xi, fix_b=True), g["gt"])
        t0 = time.time()
        H, rhs, adj, free = build_system(g, states)
        lam = lam_rel * H.diagonal().max()
        A = (H + lam * sp.identity(H.shape[0])).tocsr()
        nnz0 = H.nnz // 49
        print(f"== {label}: nb={free.size} nnzb={nnz0} lam={lam:.3g} (build {time.time()-t0:.1f}s)", flush=True)
        L0 = Level(); L0.Dinv = block_diag_inv(A, 0.0)
        t0 = time.time()
        if not os.environ.get("SKIP_BJ"):
            _, it = pcg(A, rhs, lambda r: bj(L0, r), 1e-8, 6000)
            print(f"block-Jacobi: {it} its ({time.time()-t0:.1f}s)", flush=True)
        for passes, omega, nu, gamma, over in CONFIGS:

            ADDITIVE0 = passes < 0
            passes = abs(passes)
            t0 = time.time()
            lv = build_hierarchy(H, lam, adj, states[free], passes, 6, 400, omega, rng)
            sizes = [l.A.shape[0] // 7 for l in lv]
            nnzs = [l.nnzb for l in lv]
            # work per PCG iteration in fine SpMV equivalents: CG SpMV + per level (nu-1 + 1 + nu) SpMVs
            cyc = [1.0]
            for k in range(1, len(lv)):
                cyc.append(cyc[-1] * (gamma if k >= 2 else 1))
            work = 1 + sum(c * (2 * nu) * n / nnz0 for c, n in zip(cyc[:-1], nnzs[:-1]))
            tb = time.time() - t0
            t0 = time.time()
            _, it = pcg(A, rhs, lambda r: vcycle(lv, 0, r, nu, gamma, over), 1e-8, 600)
            print(f"AMG passes={passes} omega={omega} nu={nu} gamma={gamma} over={over}: levels {sizes} nnzb {nnzs} "
                  f"-> {it} its, work/it {work:.2f} SpMV-eq, total {it*work:.0f} "
                  f"(setup {tb:.1f}s, solve {time.time()-t0:.1f}s)", flush=True)


if __name__ == "__main__":
    main()
