"""CPU experiment (numpy, oracle Jacobians, scripts/proto_amg.py): can the rejected trials of one LM iteration -- same H and b,
lambda growing 2x, 8x, 64x ... -- start from a Galerkin projection onto the Krylov basis of the first solve (or simply from
the previous solution)?  Prints PCG iterations cold / warm / projected per trial.  python scripts/proto_krylov_projection.py [V] [lambda0/mean diag]"""
import os, sys, time
import numpy as np, scipy.sparse as sp
sys.path.insert(0, "/root/repo/scripts"); sys.path.insert(0, "/root/repo")
import proto_amg as PA
from sim3opt_amd import synth, sim3np as S3
V = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
side = int(round((V / 10) ** 0.5))
g = synth.manhattan(V, 10 * V, dims=(side, side, 10))
rng = np.random.default_rng(0)
xi = rng.standard_normal((V, 7)) * np.array([1e-3] * 3 + [1e-2] * 3 + [1e-3])
states = S3.mul(S3.exp(xi, fix_b=True), g["gt"])
H, rhs, adj, free = PA.build_system(g, states)
dmax = H.diagonal().max(); dmean = H.diagonal().mean()
print("n", H.shape[0], "diag max %.3g mean %.3g" % (dmax, dmean))
def pcg(A, b, M, tol, maxit, x0=None, rz_ref=None, store=None):
    x = np.zeros_like(b) if x0 is None else x0.copy()
    r = b - A @ x if x0 is not None else b.copy()
    z = M(r); p = z.copy(); rz = r @ z
    rz0 = rz if rz_ref is None else rz_ref
    if store is not None: store.append(z.copy())
    if rz <= tol * tol * rz0: return x, 0, rz0
    for it in range(1, maxit + 1):
        q = A @ p; alpha = rz / (p @ q); x += alpha * p; r -= alpha * q
        z = M(r); rzn = r @ z
        if store is not None: store.append(z.copy())
        if rzn <= tol * tol * rz0: return x, it, rz0
        p = z + (rzn / rz) * p; rz = rzn
    return x, maxit, rz0
lam0 = float(sys.argv[2]) * dmean if len(sys.argv) > 2 else 5e-8 * dmean
def setup(lam):
    lv = PA.build_hierarchy(H, lam, adj, states[free], 3, 6, 400, 0.8, np.random.default_rng(0))
    A = (H + lam * sp.identity(H.shape[0])).tocsr()
    return A, (lambda r: PA.vcycle(lv, 0, r, 1, 2, 1.0))
A0, M0 = setup(lam0)
Z = []
x0, it0, rzref = pcg(A0, rhs, M0, 1e-8, 400, store=Z)
print("trial 0: lambda %.3g: %d its, basis %d" % (lam0, it0, len(Z)))
Vb = np.array(Z).T                      # n x K
Vb = Vb / np.linalg.norm(Vb, axis=0)
AV = H @ Vb
G1 = Vb.T @ AV; G2 = Vb.T @ Vb; gv = Vb.T @ rhs
lam = lam0; nu = 2.0
for t in range(1, 7):
    lam *= nu; nu *= 2
    A, M = setup(lam)
    _, it_cold, _ = pcg(A, rhs, M, 1e-8, 400)
    _, it_warm, _ = pcg(A, rhs, M, 1e-8, 400, x0=x0, rz_ref=rzref)
    y = np.linalg.solve(G1 + lam * G2, gv)
    xp = Vb @ y
    res = np.linalg.norm(rhs - A @ xp) / np.linalg.norm(rhs)
    _, it_proj, _ = pcg(A, rhs, M, 1e-8, 400, x0=xp, rz_ref=rzref)
    print("trial %d: lambda %.3g (%.1e x mean diag): cold %d, warm(prev x) %d, projected %d its (rel residual of projection %.2e)" % (t, lam, lam / dmean, it_cold, it_warm, it_proj, res))

# ---- deflation with the lowest Ritz vectors of the first solve's basis (Saad, Yeung, Erhel, Guyomarc'h 2000) ----
import scipy.linalg as sla
def deflated_pcg(A, b, M, W, AW, tol, maxit, rz_ref):
    G = W.T @ AW
    Gi = np.linalg.inv(G)
    x = W @ (Gi @ (W.T @ b))
    r = b - A @ x
    z = M(r)
    p = z - W @ (Gi @ (AW.T @ z))
    rz = r @ z
    if rz <= tol * tol * rz_ref: return x, 0
    for it in range(1, maxit + 1):
        q = A @ p; alpha = rz / (p @ q); x += alpha * p; r -= alpha * q
        z = M(r); rzn = r @ z
        if rzn <= tol * tol * rz_ref: return x, it
        p = z + (rzn / rz) * p - W @ (Gi @ (AW.T @ z)); rz = rzn
    return x, maxit
th, Y = sla.eigh(G1 + lam0 * G2, G2)
print("lowest Ritz values of H + lambda0 in the basis:", np.array2string(th[:10], precision=3))
for kdef in (4, 8, 16):
    W = Vb @ Y[:, :kdef]
    HW = H @ W
    lam = lam0; nu = 2.0; out = []
    for t in range(1, 6):
        lam *= nu; nu *= 2
        A, M = setup(lam)
        _, itd = deflated_pcg(A, rhs, M, W, HW + lam * W, 1e-8, 400, rzref)
        out.append(itd)
    print("deflating the %d lowest Ritz vectors: trials 1..5 need %s iterations" % (kdef, out))
