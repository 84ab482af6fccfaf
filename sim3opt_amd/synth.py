"""Synthetic Sim(3) pose-graph generators (BASELINE.json configs 2 and 3).

The reference ships no generator (its only graph is KITTI-00, kitti_surf.cpp:562-670); these
produce graphs with the reference's conventions so the same arrays feed the HIP library and
the CPU oracle:
  * estimate of vertex i is S_iw (world -> camera i), scale 1 in the ground truth
  * odometry edge: v0 = i, v1 = i-1, loop edge: v0 = a < v1 = b       (kitti_surf.cpp:624-670)
  * measurement C = exp(noise) * S_v1 * S_v0^-1, so e = log(C * S_v0 * S_v1^-1) = noise at truth
  * initial guess = dead reckoning over the odometry measurements (zero odometry residual at
    the start, like the reference's KITTI graph), vertex 0 fixed, information = I7
Host-side numpy only; nothing here runs during optimisation.
"""
import numpy as np

from . import sim3np as S3

# noise (omega, upsilon, sigma) of loop-closure measurements
LOOP_SIGMA = (0.01, 0.05, 0.01)
# odometry noise is set so that dead-reckoning drift at the end of the walk is about this
# (rotation [rad], log-scale); translation noise is 5x the rotation noise per step
DRIFT_TARGET = 0.15


def _noise(rng, n, sig):
    xi = np.empty((n, 7))
    xi[:, 0:3] = rng.standard_normal((n, 3)) * sig[0]
    xi[:, 3:6] = rng.standard_normal((n, 3)) * sig[1]
    xi[:, 6] = rng.standard_normal(n) * sig[2]
    return xi


def _measure(Sgt, v0, v1, xi):
    # exact exp (fix_b): the noise must not pass through the reference's as-written B coefficient
    return S3.mul(S3.exp(xi, fix_b=True), S3.mul(Sgt[v1], S3.inv(Sgt[v0])))


def _dead_reckon(S0, odo_meas):
    """S_i = C_i^-1 * S_{i-1} makes every odometry residual log(C_i S_i S_{i-1}^-1) zero."""
    n = odo_meas.shape[0] + 1
    out = np.empty((n, 8))
    out[0] = S0
    Cinv = S3.inv(odo_meas)
    # sequential composition; quaternion algebra inlined for speed
    q = S0[:4].copy()
    t = S0[4:7].copy()
    s = float(S0[7])
    for i in range(1, n):
        c = Cinv[i - 1]
        cq, ct, cs = c[:4], c[4:7], c[7]
        # new = c * prev
        uv = 2.0 * np.cross(cq[:3], t)
        rt = t + cq[3] * uv + np.cross(cq[:3], uv)
        t = cs * rt + ct
        q = np.array([
            cq[3] * q[0] + cq[0] * q[3] + cq[1] * q[2] - cq[2] * q[1],
            cq[3] * q[1] + cq[1] * q[3] + cq[2] * q[0] - cq[0] * q[2],
            cq[3] * q[2] + cq[2] * q[3] + cq[0] * q[1] - cq[1] * q[0],
            cq[3] * q[3] - cq[0] * q[0] - cq[1] * q[1] - cq[2] * q[2],
        ])
        s = cs * s
        out[i, :4] = q
        out[i, 4:7] = t
        out[i, 7] = s
    out[:, :4] /= np.linalg.norm(out[:, :4], axis=1, keepdims=True)
    return out


def _finish(Sgt, odo_v0, odo_v1, loop_v0, loop_v1, seed_noise, odo_sigma=None):
    V = Sgt.shape[0]
    rng = np.random.default_rng(seed_noise)
    if odo_sigma is None:
        so = DRIFT_TARGET / np.sqrt(max(V - 1, 1))
        odo_sigma = (so, 5 * so, so)
    odo_meas = _measure(Sgt, odo_v0, odo_v1, _noise(rng, odo_v0.shape[0], odo_sigma))
    loop_meas = _measure(Sgt, loop_v0, loop_v1, _noise(rng, loop_v0.shape[0], LOOP_SIGMA))
    init = _dead_reckon(Sgt[0], odo_meas)
    fixed = np.zeros(V, dtype=np.uint8)
    fixed[0] = 1
    return dict(
        states=init, fixed=fixed,
        v0=np.concatenate([loop_v0, odo_v0]).astype(np.int32),
        v1=np.concatenate([loop_v1, odo_v1]).astype(np.int32),
        meas=np.concatenate([loop_meas, odo_meas]),
        gt=Sgt, n_loop=int(loop_v0.shape[0]),
    )


def _gt_from_cam2world(R_wi, p_i):
    """S_iw = (R_wi^T, -R_wi^T p_i, 1)."""
    q_wi = S3.R_to_quat(R_wi)
    q_iw = S3.quat_conj(q_wi)
    t = S3.quat_rot(q_iw, -p_i)
    return S3.make(q_iw, t, 1.0)


def chain_loop(V=10000, E=20000, seed_graph=20240601, seed_noise=20240602, min_gap=50):
    """Config 2: random-walk chain with E-(V-1) uniformly random long-range loop closures."""
    assert E >= V - 1
    rng = np.random.default_rng(seed_graph)
    # heading change N(0, 0.1 rad) about a random axis, step U(0.5, 1.5) along the camera z axis
    axis = rng.standard_normal((V, 3))
    axis /= np.linalg.norm(axis, axis=1, keepdims=True)
    ang = rng.standard_normal(V) * 0.1
    step = rng.uniform(0.5, 1.5, V)
    dq = np.concatenate([axis * np.sin(ang / 2)[:, None], np.cos(ang / 2)[:, None]], axis=1)
    q = np.empty((V, 4))
    p = np.zeros((V, 3))
    q[0] = [0, 0, 0, 1]
    for i in range(1, V):
        q[i] = S3.quat_mul(q[i - 1], dq[i])
        q[i] /= np.linalg.norm(q[i])
        p[i] = p[i - 1] + S3.quat_rot(q[i - 1], np.array([0.0, 0.0, step[i]]))
    Sgt = _gt_from_cam2world(S3.quat_to_R(q), p)
    n_loop = E - (V - 1)
    gap = min(min_gap, max(2, V // 4))
    pairs = set()
    while len(pairs) < n_loop:
        a = rng.integers(0, V, size=2 * (n_loop - len(pairs)) + 16)
        b = rng.integers(0, V, size=a.shape[0])
        lo, hi = np.minimum(a, b), np.maximum(a, b)
        for x, y in zip(lo[hi - lo >= gap], hi[hi - lo >= gap]):
            if len(pairs) >= n_loop:
                break
            pairs.add((int(x), int(y)))
    pairs = np.array(sorted(pairs), dtype=np.int64).reshape(-1, 2)
    pairs = pairs[rng.permutation(pairs.shape[0])]
    odo_v0 = np.arange(1, V)
    odo_v1 = np.arange(0, V - 1)
    return _finish(Sgt, odo_v0, odo_v1, pairs[:, 0], pairs[:, 1], seed_noise)


def _rotation_group():
    """The 24 proper rotations of the cube, their composition table and traces."""
    mats = []
    import itertools
    for perm in itertools.permutations(range(3)):
        for signs in itertools.product([1, -1], repeat=3):
            M = np.zeros((3, 3), dtype=np.int64)
            for r in range(3):
                M[r, perm[r]] = signs[r]
            if round(np.linalg.det(M)) == 1:
                mats.append(M)
    mats = np.array(mats)
    key = {m.tobytes(): k for k, m in enumerate(mats)}
    table = np.empty((24, 24), dtype=np.int64)
    for a in range(24):
        for b in range(24):
            table[a, b] = key[(mats[a] @ mats[b]).tobytes()]
    return mats, table


def manhattan(V=100000, E=1000000, dims=(100, 100, 10), seed_graph=20240611,
              seed_noise=20240612, p_turn=0.2, radius=2, per_cell=3):
    """Config 3: 3-D Manhattan walk on a lattice; loops to earlier vertices within L1 `radius`."""
    assert E >= V - 1
    rng = np.random.default_rng(seed_graph)
    mats, table = _rotation_group()
    ident = int(np.where((mats == np.eye(3, dtype=np.int64)).all(axis=(1, 2)))[0][0])
    # 90-degree turns about the body axes (the 6 group elements with trace 1)
    turns = [k for k in range(24) if np.trace(mats[k]) == 1]
    fwd = mats[:, :, 2]  # world direction of the camera z axis for each orientation
    dims = np.array(dims)
    pos = np.empty((V, 3), dtype=np.int64)
    ori = np.empty(V, dtype=np.int64)
    pos[0] = dims // 2
    ori[0] = ident
    u = rng.random(V)
    pick = rng.integers(0, len(turns), size=(V, 8))
    for i in range(1, V):
        o = ori[i - 1]
        if u[i] < p_turn:
            o = table[o, turns[pick[i, 0]]]
        nxt = pos[i - 1] + fwd[o]
        k = 1
        while (nxt < 0).any() or (nxt >= dims).any():  # bounce: turn until the step stays inside
            o = table[o, turns[pick[i, k % 8]]]
            nxt = pos[i - 1] + fwd[o]
            k += 1
            if k > 64:
                o = table[o, turns[int(rng.integers(0, len(turns)))]]
        ori[i] = o
        pos[i] = nxt
    Sgt = _gt_from_cam2world(mats[ori].astype(np.float64), pos.astype(np.float64))
    # candidate loop pairs: vertices in lattice cells within L1 distance <= radius; the radius
    # grows (deterministically) until there are enough candidates
    cell = (pos[:, 0] * dims[1] + pos[:, 1]) * dims[2] + pos[:, 2]
    order = np.argsort(cell, kind="stable")
    cs = cell[order]
    n_loop = E - (V - 1)
    idx = np.arange(V)
    while True:
        offs = [(dx, dy, dz) for dx in range(-radius, radius + 1)
                for dy in range(-radius, radius + 1) for dz in range(-radius, radius + 1)
                if abs(dx) + abs(dy) + abs(dz) <= radius]
        cand = []
        for (dx, dy, dz) in offs:
            q = pos + np.array([dx, dy, dz])
            ok = ((q >= 0) & (q < dims)).all(axis=1)
            qc = (q[:, 0] * dims[1] + q[:, 1]) * dims[2] + q[:, 2]
            lo = np.searchsorted(cs, qc, side="left")
            hi = np.searchsorted(cs, qc, side="right")
            for k in range(per_cell):
                sel = ok & (lo + k < hi)
                j = order[np.minimum(lo + k, V - 1)]
                sel &= j < idx - 1  # earlier vertex, not the odometry predecessor
                cand.append(np.stack([j[sel], idx[sel]], axis=1))
        cand = np.unique(np.concatenate(cand), axis=0)
        # drop pairs whose relative rotation is a half turn (log degenerates at pi)
        rel = np.einsum("nij,nik->njk", mats[ori[cand[:, 0]]], mats[ori[cand[:, 1]]])
        cand = cand[np.trace(rel, axis1=1, axis2=2) >= 0]
        if cand.shape[0] >= n_loop:
            break
        radius += 1
        if radius > 8:
            raise ValueError(f"only {cand.shape[0]} loop candidates for {n_loop} loops")
    cand = cand[rng.permutation(cand.shape[0])[:n_loop]]
    odo_v0 = np.arange(1, V)
    odo_v1 = np.arange(0, V - 1)
    return _finish(Sgt, odo_v0, odo_v1, cand[:, 0], cand[:, 1], seed_noise)


def positions(states):
    """Camera centres t(S_wi) of estimates S_iw (what the reference writes, kitti_surf.cpp:691-698)."""
    return S3.inv(np.asarray(states))[:, 4:7]


def rmse(states_a, states_b):
    d = positions(states_a) - positions(states_b)
    return float(np.sqrt((d ** 2).sum(axis=1).mean()))
