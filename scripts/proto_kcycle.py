"""CPU experiment: K-cycle (Notay's AGMG: two flexible-CG steps on every coarse level, preconditioned by the next level's cycle) against
the product's stationary W-type revisits with over-correction (2/3/3, 1.8 / 1.6).  Outer solver: flexible PCG (Polak-Ribiere beta) for both.
python scripts/proto_kcycle.py [V=4000] [levels=3]"""
import os, sys, time
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proto_amg as PA
from sim3opt_amd import synth, sim3np as S3
V = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 3
side = int(round((V / 10) ** 0.5))
g = synth.manhattan(V, 10 * V, dims=(side, side, 10))
rng = np.random.default_rng(0)
count = {"l1": 0, "l2": 0}


def smooth0(L, r):
    return L.omega * PA.bj(L, r)


def vk(levels, k, r, mode, over):
    """one application of the level-k preconditioner: smoothing, coarse solve (mode: 'K' two FCG steps / ('W', visits) stationary), smoothing"""
    L = levels[k]
    if k == len(levels) - 1:
        return L.lu.solve(r)
    count["l%d" % k] = count.get("l%d" % k, 0) + 1
    x = smooth0(L, r)
    rc = L.P.T @ (r - L.A @ x)
    Lc = levels[k + 1]
    if k + 1 == len(levels) - 1:
        xc = Lc.lu.solve(rc)
    elif mode == "K":
        # two steps of flexible CG on the coarse system, preconditioned by the next level's cycle (Notay 2008, Alg. 3.2)
        c = vk(levels, k + 1, rc, mode, over)
        v = Lc.A @ c
        rho1, alpha1 = c @ v, c @ rc
        xc = (alpha1 / rho1) * c
        r2 = rc - (alpha1 / rho1) * v
        if np.linalg.norm(r2) > 0.25 * np.linalg.norm(rc):
            d = vk(levels, k + 1, r2, mode, over)
            w = Lc.A @ d
            gamma, beta, alpha2 = d @ v, d @ w, d @ r2
            rho2 = beta - gamma * gamma / rho1
            xc = (alpha1 / rho1 - gamma * alpha2 / (rho1 * rho2)) * c + (alpha2 / rho2) * d
    else:
        visits = mode[1]
        xc = vk(levels, k + 1, rc, mode, over)
        for _ in range(visits[k + 1] - 1):
            xc = xc + vk(levels, k + 1, rc - Lc.A @ xc, mode, over)
    ov = 1.0 if mode == "K" else over[0 if k == 0 else 1]
    x = x + ov * (L.P @ xc)
    return x + smooth0(L, r - L.A @ x)


def fpcg(A, b, M, tol, maxit):
    x = np.zeros_like(b); r = b.copy(); z = M(r); p = z.copy(); rz = r @ z; rz0 = rz
    for it in range(1, maxit + 1):
        q = A @ p; alpha = rz / (p @ q); x += alpha * p
        rn = r - alpha * q
        zn = M(rn)
        rzn = rn @ zn
        if abs(rzn) <= tol * tol * rz0: return x, it
        beta = (zn @ (rn - r)) / rz   # Polak-Ribiere: flexible
        p = zn + beta * p; r = rn; z = zn; rz = rzn
    return x, maxit


for label, states, lam_rel in (("initial", g["states"], 1e-5), ("near optimum", None, 1e-8)):
    if states is None:
        xi = rng.standard_normal((V, 7)) * np.array([1e-3] * 3 + [1e-2] * 3 + [1e-3])
        states = S3.mul(S3.exp(xi, fix_b=True), g["gt"])
    H, rhs, adj, free = PA.build_system(g, states)
    lam = lam_rel * H.diagonal().max()
    A = (H + lam * sp.identity(H.shape[0])).tocsr()
    lv = PA.build_hierarchy(H, lam, adj, states[free], 3, NL, 150, 0.9, np.random.default_rng(0))
    print("== %s: levels %s" % (label, [l.A.shape[0] // 7 for l in lv]), flush=True)
    for name, mode, over in (("W revisits 2/3/3, over-correction 1.8 / 1.6 (the product)", ("W", [1, 2, 3, 3, 3]), (1.8, 1.6)),
                             ("W revisits 2/3/3, no over-correction", ("W", [1, 2, 3, 3, 3]), (1.0, 1.0)),
                             ("V cycle, over-correction 1.8 / 1.6", ("W", [1, 1, 1, 1, 1]), (1.8, 1.6)),
                             ("K-cycle (two FCG steps per coarse level)", "K", (1.0, 1.0))):
        count.clear()
        _, it = fpcg(A, rhs, lambda r: vk(lv, 0, r, mode, over), 1e-8, 300)
        print("  %-62s %3d outer iterations; level visits per iteration %s" % (name, it, {k: round(v / it, 1) for k, v in sorted(count.items())}), flush=True)
