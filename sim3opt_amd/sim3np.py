"""Vectorised numpy Sim(3) helpers for HOST-SIDE data preparation only.

Used by the synthetic graph generators (synth.py), the KITTI graph builder
(kitti.py) and the tests.  Nothing here is on the optimisation path: the LM
inner loop runs exclusively in the HIP library (csrc/).

State layout everywhere: 8 doubles [qx qy qz qw tx ty tz s]  (Eigen coeffs()
order, as printed by the reference at kitti_surf.cpp:698).
Tangent order: [omega(3), upsilon(3), sigma]  (g2o convention).
Formulae follow the in-tree authority sim3_rv.h:125-190 (exp), :242-320 (ln),
:199-220 (inverse / compose).
"""
import numpy as np

EPS = 1e-5
# Default of the `fix_b` argument of exp/log below.  0: small-theta B coefficient as written in
# sim3_rv.h:166/:290 (reference behaviour, B ~ 1/sigma^3); 1: exact limit.  Data generation
# (synth.py) always passes fix_b=True: with the as-written B a noise vector with theta < 1e-5
# and |sigma| > 1e-5 would get its translation multiplied by ~1e3 and corrupt the graph.
FIX_SMALL_ANGLE_B = 0


def quat_mul(a, b):
    ax, ay, az, aw = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bx, by, bz, bw = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by + ay * bw + az * bx - ax * bz,
        aw * bz + az * bw + ax * by - ay * bx,
        aw * bw - ax * bx - ay * by - az * bz,
    ], axis=-1)


def quat_conj(q):
    return q * np.array([-1.0, -1.0, -1.0, 1.0])


def quat_rot(q, v):
    qv = q[..., :3]
    uv = 2.0 * np.cross(qv, v)
    return v + q[..., 3:4] * uv + np.cross(qv, uv)


def quat_to_R(q):
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - 2 * (y * y + z * z)
    R[..., 0, 1] = 2 * (x * y - z * w)
    R[..., 0, 2] = 2 * (x * z + y * w)
    R[..., 1, 0] = 2 * (x * y + z * w)
    R[..., 1, 1] = 1 - 2 * (x * x + z * z)
    R[..., 1, 2] = 2 * (y * z - x * w)
    R[..., 2, 0] = 2 * (x * z - y * w)
    R[..., 2, 1] = 2 * (y * z + x * w)
    R[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def R_to_quat(R):
    """Batch matrix -> quaternion (xyzw), numerically safe on all branches."""
    R = np.asarray(R, dtype=np.float64)
    shp = R.shape[:-2]
    Rf = R.reshape(-1, 3, 3)
    q = np.empty((Rf.shape[0], 4))
    tr = Rf[:, 0, 0] + Rf[:, 1, 1] + Rf[:, 2, 2]
    pos = tr > 0
    if pos.any():
        M = Rf[pos]
        t = np.sqrt(tr[pos] + 1.0)
        w = 0.5 * t
        t = 0.5 / t
        q[pos] = np.stack([(M[:, 2, 1] - M[:, 1, 2]) * t, (M[:, 0, 2] - M[:, 2, 0]) * t,
                           (M[:, 1, 0] - M[:, 0, 1]) * t, w], axis=-1)
    for idx in np.nonzero(~pos)[0]:
        M = Rf[idx]
        i = 0
        if M[1, 1] > M[0, 0]:
            i = 1
        if M[2, 2] > M[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(M[i, i] - M[j, j] - M[k, k] + 1.0)
        qq = np.empty(4)
        qq[i] = 0.5 * t
        t = 0.5 / t
        qq[3] = (M[k, j] - M[j, k]) * t
        qq[j] = (M[j, i] + M[i, j]) * t
        qq[k] = (M[k, i] + M[i, k]) * t
        q[idx] = qq
    return q.reshape(shp + (4,))


def euler_rpy_to_R(r, p, y):
    """kittiDetector.h:225-243: R = Rz(yaw) Ry(pitch) Rx(roll)."""
    cr, sr, cp, sp, ch, sh = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([
        [cp * ch, sp * sr * ch - cr * sh, cr * sp * ch + sh * sr],
        [cp * sh, sr * sp * sh + cr * ch, cr * sp * sh - sr * ch],
        [-sp, sr * cp, cr * cp],
    ])


def make(q, t, s):
    q = np.asarray(q, dtype=np.float64)
    t = np.asarray(t, dtype=np.float64)
    s = np.broadcast_to(np.asarray(s, dtype=np.float64), q.shape[:-1])
    return np.concatenate([q, t, s[..., None]], axis=-1)


def mul(a, b):
    q = quat_mul(a[..., :4], b[..., :4])
    t = a[..., 7:8] * quat_rot(a[..., :4], b[..., 4:7]) + a[..., 4:7]
    return np.concatenate([q, t, a[..., 7:8] * b[..., 7:8]], axis=-1)


def inv(a):
    qc = quat_conj(a[..., :4])
    t = quat_rot(qc, (-1.0 / a[..., 7:8]) * a[..., 4:7])
    return np.concatenate([qc, t, 1.0 / a[..., 7:8]], axis=-1)


def _skew(w):
    W = np.zeros(w.shape[:-1] + (3, 3))
    W[..., 0, 1] = -w[..., 2]
    W[..., 0, 2] = w[..., 1]
    W[..., 1, 0] = w[..., 2]
    W[..., 1, 2] = -w[..., 0]
    W[..., 2, 0] = -w[..., 1]
    W[..., 2, 1] = w[..., 0]
    return W


def _abc(sigma, s, theta, small_theta, fix_b):
    small_sigma = np.abs(sigma) < EPS
    th = np.where(small_theta, 1.0, theta)
    sg = np.where(small_sigma, 1.0, sigma)
    A0 = np.where(small_theta, 0.5, (1 - np.cos(th)) / th ** 2)
    B0 = np.where(small_theta, 1.0 / 6.0, (th - np.sin(th)) / th ** 3)
    C1 = (s - 1) / sg
    A1s = ((sg - 1) * s + 1) / sg ** 2
    B1s = ((0.5 * sg ** 2 - sg + 1) * s - (1.0 if fix_b else 0.0)) / sg ** 3
    a, b, c = s * np.sin(th), s * np.cos(th), th ** 2 + sg ** 2
    A1 = (a * sg + (1 - b) * th) / (th * c)
    B1 = (C1 - ((b - 1) * sg + a * th) / c) / th ** 2
    A = np.where(small_sigma, A0, np.where(small_theta, A1s, A1))
    B = np.where(small_sigma, B0, np.where(small_theta, B1s, B1))
    Cc = np.where(small_sigma, 1.0, C1)
    return A, B, Cc


def exp(xi, fix_b=None):
    fix_b = FIX_SMALL_ANGLE_B if fix_b is None else fix_b
    xi = np.asarray(xi, dtype=np.float64)
    om, up, sigma = xi[..., :3], xi[..., 3:6], xi[..., 6]
    theta = np.linalg.norm(om, axis=-1)
    small = theta < EPS
    Om = _skew(om)
    Om2 = Om @ Om
    s = np.exp(sigma)
    A, B, Cc = _abc(sigma, s, theta, small, fix_b)
    th = np.where(small, 1.0, theta)
    k1 = np.where(small, 1.0, np.sin(th) / th)
    k2 = np.where(small, 1.0, (1 - np.cos(th)) / th ** 2)
    I = np.eye(3)
    R = I + k1[..., None, None] * Om + k2[..., None, None] * Om2
    W = A[..., None, None] * Om + B[..., None, None] * Om2 + Cc[..., None, None] * I
    t = np.einsum("...ij,...j->...i", W, up)
    return make(R_to_quat(R), t, s)


def log(S, fix_b=None):
    fix_b = FIX_SMALL_ANGLE_B if fix_b is None else fix_b
    S = np.asarray(S, dtype=np.float64)
    s = S[..., 7]
    sigma = np.log(s)
    R = quat_to_R(S[..., :4])
    d = 0.5 * (R[..., 0, 0] + R[..., 1, 1] + R[..., 2, 2] - 1)
    dR = np.stack([R[..., 2, 1] - R[..., 1, 2], R[..., 0, 2] - R[..., 2, 0],
                   R[..., 1, 0] - R[..., 0, 1]], axis=-1)
    small = d > 1 - EPS
    dc = np.where(small, 0.0, np.clip(d, -1.0, 1.0))
    theta = np.where(small, 0.0, np.arccos(dc))
    k = np.where(small, 0.5, theta / (2 * np.sqrt(np.maximum(1 - dc * dc, 1e-300))))
    om = k[..., None] * dR
    A, B, Cc = _abc(sigma, s, theta, small, fix_b)
    Om = _skew(om)
    W = A[..., None, None] * Om + B[..., None, None] * (Om @ Om) + Cc[..., None, None] * np.eye(3)
    up = np.linalg.solve(W, S[..., 4:7, None])[..., 0]
    return np.concatenate([om, up, sigma[..., None]], axis=-1)


def edge_error(C, S0, S1, fix_b=None):
    """EdgeSim3::computeError: log(C * S0 * S1^-1)."""
    return log(mul(mul(C, S0), inv(S1)), fix_b)


def identity(n=None):
    one = np.array([0, 0, 0, 1, 0, 0, 0, 1], dtype=np.float64)
    return one if n is None else np.tile(one, (n, 1))
