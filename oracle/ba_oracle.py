"""CPU restatement (numpy / scipy) of the reference's `ba_demo` path (bal_example.cpp:44-243).

TEST INFRASTRUCTURE ONLY -- imported by tests/ only; the product (sim3opt_amd/) never imports this.

PARITY UNPINNED: the arithmetic lives in g2o @ 8564e1e (types/sba: VertexSE3Expmap,
VertexSBAPointXYZ, EdgeProjectXYZ2UV, CameraParameters; core: OptimizationAlgorithmLevenberg,
BlockSolver_6_3 with Schur complement, RobustKernelHuber; solvers/eigen), which is not under
/root/reference and cannot be fetched; the reference stores no BA output.  This file restates the
published behaviour [upstream-recall]; what pins it is in tests/test_ba.py: analytic Jacobians
against central differences, the BAL conventions of the reference's own writer
(drawPTAMPoints.cpp:218-283) and reader (bal_example.cpp:104-189), scipy's sparse LU.

What ba_demo does: cameras are g2o::VertexSE3Expmap (T_w2c as unit quaternion + translation, update
T <- exp([omega, upsilon]) T), points g2o::VertexSBAPointXYZ (marginalised), every observation a
g2o::EdgeProjectXYZ2UV with information I / PIXEL_NOISE^2, RobustKernelHuber(2.5), ONE shared
CameraParameters(718.856, (607.1928, 185.2157), 0) -- the focal length / distortion columns of the BAL
file are read and ignored (bal_example.cpp:90-97, :160-175); LM, 5 iterations by default (:53).
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


# ---- quaternion helpers on (n, 4) arrays, Eigen coeffs() order x y z w ----
def quat_to_R(q):
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - 2 * (y * y + z * z); R[..., 0, 1] = 2 * (x * y - z * w); R[..., 0, 2] = 2 * (x * z + y * w)
    R[..., 1, 0] = 2 * (x * y + z * w); R[..., 1, 1] = 1 - 2 * (x * x + z * z); R[..., 1, 2] = 2 * (y * z - x * w)
    R[..., 2, 0] = 2 * (x * z - y * w); R[..., 2, 1] = 2 * (y * z + x * w); R[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def R_to_quat(R):
    """Eigen's Quaternion(Matrix3): trace branch, else the largest diagonal entry."""
    R = np.asarray(R)
    out = np.empty(R.shape[:-2] + (4,))
    flat = R.reshape(-1, 3, 3)
    o = out.reshape(-1, 4)
    for n, M in enumerate(flat):
        tr = M[0, 0] + M[1, 1] + M[2, 2]
        if tr > 0:
            k = np.sqrt(tr + 1.0)
            w = 0.5 * k
            k = 0.5 / k
            o[n] = [(M[2, 1] - M[1, 2]) * k, (M[0, 2] - M[2, 0]) * k, (M[1, 0] - M[0, 1]) * k, w]
        else:
            i = 0
            if M[1, 1] > M[0, 0]:
                i = 1
            if M[2, 2] > M[i, i]:
                i = 2
            j, l = (i + 1) % 3, (i + 2) % 3
            k = np.sqrt(M[i, i] - M[j, j] - M[l, l] + 1.0)
            q = np.zeros(4)
            q[i] = 0.5 * k
            k = 0.5 / k
            q[3] = (M[l, j] - M[j, l]) * k
            q[j] = (M[j, i] + M[i, j]) * k
            q[l] = (M[l, i] + M[i, l]) * k
            o[n] = q
    return out


def angle_axis_to_quat(aa):
    """ceres AngleAxisToQuaternion (called at bal_example.cpp:168), returns x y z w."""
    aa = np.asarray(aa, dtype=np.float64)
    th2 = (aa * aa).sum(-1)
    th = np.sqrt(th2)
    small = th2 <= 0.0
    k = np.where(small, 0.5, np.sin(0.5 * np.where(small, 1.0, th)) / np.where(small, 1.0, th))
    w = np.where(small, 1.0, np.cos(0.5 * th))
    return np.concatenate([aa * k[..., None], w[..., None]], axis=-1)


def skew(w):
    W = np.zeros(w.shape[:-1] + (3, 3))
    W[..., 0, 1] = -w[..., 2]; W[..., 0, 2] = w[..., 1]
    W[..., 1, 0] = w[..., 2]; W[..., 1, 2] = -w[..., 0]
    W[..., 2, 0] = -w[..., 1]; W[..., 2, 1] = w[..., 0]
    return W


def se3_exp(upd):
    """g2o::SE3Quat::exp, update = [omega(3), upsilon(3)]: returns (R, t)."""
    om, up = upd[..., :3], upd[..., 3:]
    th = np.linalg.norm(om, axis=-1)
    Om = skew(om)
    Om2 = Om @ Om
    small = th < 1e-5
    t1 = np.where(small, 1.0, th)
    a = np.where(small, 1.0, np.sin(t1) / t1)
    b = np.where(small, 1.0, (1 - np.cos(t1)) / t1 ** 2)
    c = np.where(small, 1.0, (t1 - np.sin(t1)) / t1 ** 3)
    I = np.eye(3)
    R = I + a[..., None, None] * Om + b[..., None, None] * Om2
    # small angle: V = R (se3quat.h); else V = I + (1-cos)/th^2 Om + (th-sin)/th^3 Om^2
    V = np.where(small[..., None, None], R, I + b[..., None, None] * Om + c[..., None, None] * Om2)
    return R, np.einsum("...ij,...j->...i", V, up)


class Problem:
    """cams: (n, 7) [qx qy qz qw tx ty tz] = T_w2c; points (m, 3); observations (cam, point, u, v)."""

    def __init__(self, cams, points, obs_cam, obs_point, obs_uv, focal=718.856, cx=607.1928, cy=185.2157,
                 huber=2.5, pixel_noise=1.0):
        self.cams = np.array(cams, dtype=np.float64).reshape(-1, 7)
        self.points = np.array(points, dtype=np.float64).reshape(-1, 3)
        self.oc = np.asarray(obs_cam, dtype=np.int64)
        self.op = np.asarray(obs_point, dtype=np.int64)
        self.uv = np.asarray(obs_uv, dtype=np.float64).reshape(-1, 2)
        self.f, self.cx, self.cy = float(focal), float(cx), float(cy)
        self.huber, self.omega = float(huber), 1.0 / float(pixel_noise) ** 2
        self.fixed = np.zeros(len(self.cams), dtype=bool)  # Vertex::setFixed on cameras

    # EdgeProjectXYZ2UV::computeError: obs - cam_map(T.map(p))
    def _camera_frame(self, cams, points):
        R = quat_to_R(cams[self.oc, :4])
        return R, np.einsum("nij,nj->ni", R, points[self.op]) + cams[self.oc, 4:7]

    def residuals(self, cams=None, points=None):
        cams = self.cams if cams is None else cams
        points = self.points if points is None else points
        _, X = self._camera_frame(cams, points)
        proj = np.stack([self.f * X[:, 0] / X[:, 2] + self.cx, self.f * X[:, 1] / X[:, 2] + self.cy], axis=1)
        return self.uv - proj

    def jacobians(self, cams=None, points=None):
        """(J_point (n, 2, 3), J_cam (n, 2, 6)) of EdgeProjectXYZ2UV::linearizeOplus (analytic)."""
        cams = self.cams if cams is None else cams
        points = self.points if points is None else points
        R, X = self._camera_frame(cams, points)
        x, y, z = X[:, 0], X[:, 1], X[:, 2]
        f, z2 = self.f, X[:, 2] ** 2
        tmp = np.zeros((len(x), 2, 3))
        tmp[:, 0, 0] = f; tmp[:, 0, 2] = -x / z * f
        tmp[:, 1, 1] = f; tmp[:, 1, 2] = -y / z * f
        Jp = (-1.0 / z)[:, None, None] * (tmp @ R)
        Jc = np.empty((len(x), 2, 6))
        Jc[:, 0, 0] = x * y / z2 * f; Jc[:, 0, 1] = -(1 + x * x / z2) * f; Jc[:, 0, 2] = y / z * f
        Jc[:, 0, 3] = -1.0 / z * f; Jc[:, 0, 4] = 0.0; Jc[:, 0, 5] = x / z2 * f
        Jc[:, 1, 0] = (1 + y * y / z2) * f; Jc[:, 1, 1] = -x * y / z2 * f; Jc[:, 1, 2] = -x / z * f
        Jc[:, 1, 3] = 0.0; Jc[:, 1, 4] = -1.0 / z * f; Jc[:, 1, 5] = y / z2 * f
        return Jp, Jc

    def robust(self, e):
        """RobustKernelHuber on e2 = e^T Omega e: (rho, weight rho')."""
        e2 = self.omega * (e * e).sum(1)
        if self.huber <= 0:
            return e2, np.ones_like(e2)
        d = self.huber
        sq = np.sqrt(e2)
        inl = e2 <= d * d
        rho = np.where(inl, e2, 2 * sq * d - d * d)
        w = np.where(inl, 1.0, d / np.where(inl, 1.0, sq))
        return rho, w

    def chi2(self, cams=None, points=None):
        return float(self.robust(self.residuals(cams, points))[0].sum())

    def system(self, cams=None, points=None):
        """Sparse normal equations over [cameras (6 each), points (3 each)]: H (CSC), b, chi2."""
        cams = self.cams if cams is None else cams
        points = self.points if points is None else points
        nc, npt = cams.shape[0], points.shape[0]
        e = self.residuals(cams, points)
        Jp, Jc = self.jacobians(cams, points)
        rho, w = self.robust(e)
        W = w * self.omega
        rows, cols, vals = [], [], []

        def add(r0, c0, M):  # M (n, a, b) blocks at scalar offsets r0, c0 (n,)
            a, bb = M.shape[1:]
            rr = r0[:, None, None] + np.arange(a)[None, :, None]
            cc = c0[:, None, None] + np.arange(bb)[None, None, :]
            rows.append(np.broadcast_to(rr, M.shape).ravel())
            cols.append(np.broadcast_to(cc, M.shape).ravel())
            vals.append(M.ravel())

        oc6, op3 = 6 * self.oc, 6 * nc + 3 * self.op
        JcW = Jc * W[:, None, None]
        JpW = Jp * W[:, None, None]
        add(oc6, oc6, np.einsum("nri,nrj->nij", JcW, Jc))
        add(op3, op3, np.einsum("nri,nrj->nij", JpW, Jp))
        Hcp = np.einsum("nri,nrj->nij", JcW, Jp)
        add(oc6, op3, Hcp)
        add(op3, oc6, Hcp.transpose(0, 2, 1))
        n = 6 * nc + 3 * npt
        b = np.zeros(n)
        np.add.at(b, oc6[:, None] + np.arange(6), -np.einsum("nri,nr->ni", JcW, e))
        np.add.at(b, op3[:, None] + np.arange(3), -np.einsum("nri,nr->ni", JpW, e))
        H = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)).tocsc()
        if self.fixed.any():  # fixed cameras leave the system: identity rows, zero right-hand side
            keep = np.ones(n)
            keep[(6 * np.where(self.fixed)[0][:, None] + np.arange(6)).ravel()] = 0.0
            D = sp.diags(keep)
            H = (D @ H @ D + sp.diags(1.0 - keep)).tocsc()
            b = b * keep
        return H, b, float(rho.sum())

    def apply(self, cams, points, dx):
        nc = cams.shape[0]
        R, t = se3_exp(dx[:6 * nc].reshape(nc, 6))
        Rn = R @ quat_to_R(cams[:, :4])
        tn = np.einsum("nij,nj->ni", R, cams[:, 4:7]) + t
        q = R_to_quat(Rn)
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        new = np.concatenate([q, tn], axis=1)
        new[self.fixed] = cams[self.fixed]
        return new, points + dx[6 * nc:].reshape(-1, 3)

    def solve(self, H, b, lam, schur=False):
        """(H + lam I) dx = b.  schur=False: one sparse LU of the whole system (the independent check
        of the GPU's Schur path); schur=True: BlockSolver_6_3's way -- eliminate the points (3x3 blocks),
        solve the reduced camera system, back-substitute -- for problems of the KITTI map's size."""
        n = H.shape[0]
        if not schur:
            return spla.spsolve(H + lam * sp.identity(n, format="csc"), b)
        n6 = 6 * self.cams.shape[0]
        Hl = (H + lam * sp.identity(n, format="csc")).tocsr()
        A, B, D = Hl[:n6, :n6], Hl[:n6, n6:], Hl[n6:, n6:].tobsr(blocksize=(3, 3))
        assert np.array_equal(D.indices, np.arange(D.shape[0] // 3))  # block diagonal
        Dinv = sp.bsr_matrix((np.linalg.inv(D.data), D.indices, D.indptr), shape=D.shape).tocsr()
        BD = B @ Dinv
        dxc = spla.spsolve((A - BD @ B.T).tocsc(), b[:n6] - BD @ b[n6:])
        return np.concatenate([dxc, Dinv @ (b[n6:] - B.T @ dxc)])

    def optimize(self, iters, tau=1e-5, max_trials=10, lam0=0.0, schur=False):
        """OptimizationAlgorithmLevenberg on the BA graph; updates self in place."""
        trace = []
        lam = lam0 if lam0 > 0 else None
        for it in range(iters):
            H, b, chi_cur = self.system()
            if lam is None:
                free = np.ones(H.shape[0], dtype=bool)
                free[(6 * np.where(self.fixed)[0][:, None] + np.arange(6)).ravel()] = False
                lam = tau * float(np.abs(H.diagonal()[free]).max())
            ni, q, rho = 2.0, 0, 0.0
            while True:
                dx = self.solve(H, b, lam, schur)
                cn, pn = self.apply(self.cams, self.points, dx)
                chi_new = self.chi2(cn, pn)
                scale = float(dx @ (lam * dx + b)) + 1e-3
                rho = (chi_cur - chi_new) / scale
                if rho > 0 and np.isfinite(chi_new):
                    lam *= max(1.0 / 3.0, min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0))
                    ni = 2.0
                    self.cams, self.points, chi_cur = cn, pn, chi_new
                else:
                    lam *= ni
                    ni *= 2.0
                q += 1
                if not (rho < 0 and q < max_trials):
                    break
            trace.append(dict(chi2=chi_cur, lam=lam, trials=q, rho=rho))
            if q == max_trials or rho == 0 or not np.isfinite(lam):
                break
        return trace


def read_bal(path, **kw):
    """The BAL file as ba_demo reads it (bal_example.cpp:104-189): focal / distortion columns ignored."""
    tok = open(path).read().split()
    nc, npt, no = int(tok[0]), int(tok[1]), int(tok[2])
    o = 3
    obs = np.array(tok[o:o + 4 * no], dtype=np.float64).reshape(no, 4); o += 4 * no
    cam9 = np.array(tok[o:o + 9 * nc], dtype=np.float64).reshape(nc, 9); o += 9 * nc
    pts = np.array(tok[o:o + 3 * npt], dtype=np.float64).reshape(npt, 3)
    cams = np.concatenate([angle_axis_to_quat(cam9[:, :3]), cam9[:, 3:6]], axis=1)
    return Problem(cams, pts, obs[:, 0].astype(np.int64), obs[:, 1].astype(np.int64), obs[:, 2:], **kw)


def synthetic(n_cams=12, n_points=400, seed=0, noise_px=0.5, outliers=0.02, f=718.856, cx=607.1928, cy=185.2157):
    """A small KITTI-like problem: cameras along a gently turning street, points ahead of them."""
    rng = np.random.default_rng(seed)
    pts = np.stack([rng.uniform(-15, 15, n_points), rng.uniform(-3, 3, n_points),
                    rng.uniform(8, 8 + 2.0 * n_cams + 40, n_points)], axis=1)
    cams = np.zeros((n_cams, 7))
    oc, op, uv = [], [], []
    for c in range(n_cams):
        yaw = 0.02 * c
        Rc2w = np.array([[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0], [-np.sin(yaw), 0, np.cos(yaw)]])
        pc = np.array([0.3 * np.sin(0.3 * c), 0.0, 2.0 * c])
        Rw2c = Rc2w.T
        tw2c = -Rw2c @ pc
        cams[c, :4] = R_to_quat(Rw2c)
        cams[c, 4:] = tw2c
        X = pts @ Rw2c.T + tw2c
        u = f * X[:, 0] / X[:, 2] + cx
        v = f * X[:, 1] / X[:, 2] + cy
        vis = (X[:, 2] > 4) & (X[:, 2] < 60) & (u > 0) & (u < 1241) & (v > 0) & (v < 376)
        idx = np.where(vis)[0]
        oc += [c] * len(idx)
        op += idx.tolist()
        uv += np.stack([u[idx], v[idx]], axis=1).tolist()
    oc, op, uv = np.array(oc), np.array(op), np.array(uv)
    # keep points seen at least twice, compact their ids
    cnt = np.bincount(op, minlength=n_points)
    keep = cnt[op] >= 2
    oc, op, uv = oc[keep], op[keep], uv[keep]
    ids = np.unique(op)
    remap = -np.ones(n_points, dtype=np.int64)
    remap[ids] = np.arange(len(ids))
    op = remap[op]
    pts = pts[ids]
    uv = uv + rng.standard_normal(uv.shape) * noise_px
    bad = rng.random(len(uv)) < outliers
    uv[bad] += rng.standard_normal((int(bad.sum()), 2)) * 40.0
    # perturbed start
    cams0 = cams.copy()
    cams0[:, 4:] += rng.standard_normal((n_cams, 3)) * 0.05
    dq = np.concatenate([rng.standard_normal((n_cams, 3)) * 0.003, np.ones((n_cams, 1))], axis=1)
    dq /= np.linalg.norm(dq, axis=1, keepdims=True)
    x1, y1, z1, w1 = dq.T
    x2, y2, z2, w2 = cams[:, :4].T
    cams0[:, :4] = np.stack([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                             w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2, w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2], axis=1)
    pts0 = pts + rng.standard_normal(pts.shape) * 0.2
    return dict(cams=cams0, points=pts0, obs_cam=oc, obs_point=op, obs_uv=uv, cams_gt=cams, points_gt=pts)
