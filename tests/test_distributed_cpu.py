"""world_size-2 and -8 gloo tests of the row-partitioned PCG (CPU).

The HIP kernels cannot run here, so this test executes the distributed ALGORITHM the engine uses
(Engine::pcg in csrc/engine.hip, multi-GPU branch) in numpy, with the engine's own partition
function (sim3opt_partition_rows_equal through the C-ABI) and the same collectives in the same order --
all-gather of z and one 2-double all-reduce of (w.z, r.z) per iteration -- over torch.distributed gloo,
and checks it against a serial solve.  A second test does the same for the multigrid-preconditioned
PCG (Engine::amg_apply, multi-GPU branch): level 0 row-partitioned, Galerkin products and restricted
residuals all-reduced, two all-gathers per iteration, the coarse level replicated -- with the library's
own aggregation (sim3opt_amg_hierarchy, host code).  The real kernels run the same paths with 2 and 3
processes on one GPU in tests/test_distributed_gpu.py.
"""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

import dist_helpers as H


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(fn, world, out):
    """mp.spawn; one retry on a fresh port only when the TCP rendezvous itself lost a race."""
    H.spawn_with_port_retry(
        lambda: mp.spawn(fn, args=(world, _free_port(), out), nprocs=world, join=True))


def _block_system(seed=0):
    from oracle import oracle as O
    from sim3opt_amd import synth
    synth.DRIFT_TARGET = 0.05
    g = synth.manhattan(60, 400, dims=(4, 4, 3), per_cell=4, seed_graph=300 + seed,
                        seed_noise=400 + seed)
    G = O.Graph(g["states"], g["fixed"], g["v0"], g["v1"], g["meas"])
    H, b = G.build_dense(O.default_options(fix_small_angle_b=1, fd_delta=1e-6))
    return H, b


def _rowptr_of(H):
    nb = H.shape[0] // 7
    nz = np.array([[np.any(H[7 * i:7 * i + 7, 7 * j:7 * j + 7] != 0) for j in range(nb)]
                   for i in range(nb)])
    return np.concatenate([[0], np.cumsum(nz.sum(1))]).astype(np.int32)


def _worker(rank, world, port, out):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import dist_helpers as D
    from sim3opt_amd import lib as L
    D.init(rank, world, port)
    H, b = _block_system()
    n = H.shape[0]
    lam = 1e-5 * np.abs(np.diag(H)).max()
    A = H + lam * np.eye(n)
    beg = L.partition_rows_equal(n // 7, world)  # the engine's rank partition
    offs = (7 * beg).astype(np.int64)
    lo, hi = offs[rank], offs[rank + 1]
    Minv = np.zeros((n, n))
    for i in range(n // 7):
        Minv[7 * i:7 * i + 7, 7 * i:7 * i + 7] = np.linalg.inv(A[7 * i:7 * i + 7, 7 * i:7 * i + 7])
    # local state; single-reduction PCG exactly as k_pcg_step does it
    x = np.zeros(n)
    r = np.zeros(n)
    r[lo:hi] = b[lo:hi]
    z = np.zeros(n)
    z[lo:hi] = Minv[lo:hi, lo:hi] @ r[lo:hi]
    p = np.zeros(n)
    sv = np.zeros(n)
    D.allgatherv(z, offs, rank)              # the SpMV gathers z from every rank
    it, gamma_old, alpha_old, gamma0 = 0, 0.0, 0.0, 0.0
    while it < 2000:
        w = A[lo:hi, :] @ z                   # this rank's block rows only
        s = np.array([w @ z[lo:hi], r[lo:hi] @ z[lo:hi]])
        D.allreduce(s, 0)                     # ONE 2-double all-reduce per iteration
        delta, gamma = s
        if it == 0:
            gamma0 = gamma
        if gamma <= 1e-24 * gamma0:
            break
        beta = 0.0 if it == 0 else gamma / gamma_old
        alpha = gamma / (delta if it == 0 else delta - beta * gamma / alpha_old)
        p[lo:hi] = z[lo:hi] + beta * p[lo:hi]
        sv[lo:hi] = w + beta * sv[lo:hi]
        x[lo:hi] += alpha * p[lo:hi]
        r[lo:hi] -= alpha * sv[lo:hi]
        z[lo:hi] = Minv[lo:hi, lo:hi] @ r[lo:hi]
        D.allgatherv(z, offs, rank)
        gamma_old, alpha_old = gamma, alpha
        it += 1
    D.allgatherv(x, offs, rank)
    m = np.array([float(it)])
    D.allreduce(m, 1)  # max: every rank ran the same number of iterations
    assert m[0] == it
    if rank == 0:
        np.save(out, x)


@pytest.mark.parametrize("world", [2, 8])
def test_partitioned_pcg_over_gloo_matches_serial(tmp_path, world):
    out = str(tmp_path / "x.npy")
    _spawn(_worker, world, out)
    x = np.load(out)
    H, b = _block_system()
    lam = 1e-5 * np.abs(np.diag(H)).max()
    xd = np.linalg.solve(H + lam * np.eye(H.shape[0]), b)
    assert np.abs(x - xd).max() < 1e-8 * np.abs(xd).max()


def test_partition_is_exhaustive_and_disjoint():
    from sim3opt_amd import lib as L
    H, _ = _block_system(1)
    rp = _rowptr_of(H)
    for world in (2, 3, 5):
        eq = L.partition_rows_equal(len(rp) - 1, world)
        assert eq[0] == 0 and eq[-1] == len(rp) - 1 and np.all(np.diff(eq) >= 0)
        assert len(set(np.diff(eq)[:-1].tolist())) <= 1  # equal spans, short tail
        beg = L.partition_rows(rp, world)
        rows = np.concatenate([np.arange(beg[r], beg[r + 1]) for r in range(world)])
        assert np.array_equal(rows, np.arange(len(rp) - 1))


# ---------------------------------------------------------------------------------------------
# the multigrid-preconditioned PCG, row-partitioned (Engine::amg_apply, multi-GPU branch): level 0
# partitioned, Galerkin products and restricted residuals all-reduced, coarse level replicated
# ---------------------------------------------------------------------------------------------
def _adjoint(S):
    """Ad(S_v) (n, 7, 7), tangent order [omega, upsilon, sigma] (amg_kernels.hpp: k_amg_adjoint)."""
    from sim3opt_amd import sim3np as S3
    R = S3.quat_to_R(S[:, :4])
    t, s = S[:, 4:7], S[:, 7]
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


def _amg_system():
    from oracle import oracle as O
    from sim3opt_amd import lib as L, synth
    synth.DRIFT_TARGET = 0.05
    g = synth.manhattan(400, 4000, dims=(6, 6, 10))
    H, b = O.Graph(g["states"], g["fixed"], g["v0"], g["v1"], g["meas"]).build_dense(
        O.default_options(fix_small_angle_b=1, fd_delta=1e-6))
    G = L.Graph()
    G.add_vertices(g["states"], g["fixed"])
    G.add_edges(g["v0"], g["v1"], g["meas"])
    rows, _, agg = G.amg_hierarchy()  # the library's own aggregation (host code, no GPU)
    nb = rows[0]
    assert len(rows) == 2 and rows[1] <= 256
    free = np.where(g["fixed"] == 0)[0]
    Ad = _adjoint(g["states"][free])
    P = np.zeros((7 * nb, 7 * rows[1]))
    for i in range(nb):
        P[7 * i:7 * i + 7, 7 * agg[i]:7 * agg[i] + 7] = Ad[i]
    return H, b, P


def _amg_pcg(A, b, P, lam, lo, hi, offs, rank, coll):
    """Single-reduction PCG with the two-level multiplicative cycle; rows [lo, hi) are this rank's.
    coll = (allreduce, allgatherv) or None for a serial run (lo = 0, hi = n)."""
    n = A.shape[0]
    omega = 0.9
    own = slice(lo, hi)
    Dinv = np.zeros((n, n))
    for i in range(lo // 7, hi // 7):
        Dinv[7 * i:7 * i + 7, 7 * i:7 * i + 7] = omega * np.linalg.inv(A[7 * i:7 * i + 7, 7 * i:7 * i + 7])
    # Galerkin product: a rank holds its own block rows, so its product is a partial sum
    Ac = P[own].T @ (A[own] - lam * np.eye(n)[own]) @ P
    if coll:
        coll[0](Ac.reshape(-1), 0)
    Ainv_c = np.linalg.inv(Ac + lam * (P.T @ P))  # damping carried as lambda W, W = P^T P (replicated)

    def gather(v):
        if coll:
            coll[1](v, offs, rank)

    def precond(r, z0):  # z0 = omega D^-1 r on own rows (what the PCG step hands over)
        x = z0.copy()
        gather(x)
        t = r[own] - A[own] @ x
        r1 = P[own].T @ t
        if coll:
            coll[0](r1, 0)
        x = x + P @ (Ainv_c @ r1)          # every rank prolongs all rows: no collective
        out = np.zeros(n)
        out[own] = x[own] + Dinv[own, own] @ (r[own] - A[own] @ x)
        gather(out)
        return out

    x = np.zeros(n); r = np.zeros(n); p = np.zeros(n); sv = np.zeros(n)
    r[own] = b[own]
    z0 = np.zeros(n); z0[own] = Dinv[own, own] @ r[own]
    z = precond(r, z0)
    it, gamma_old, alpha_old, gamma0 = 0, 0.0, 0.0, 0.0
    while it < 500:
        w = A[own] @ z
        s = np.array([w @ z[own], r[own] @ z[own]])
        if coll:
            coll[0](s, 0)
        delta, gamma = s
        if it == 0:
            gamma0 = gamma
        if gamma <= 1e-24 * gamma0:
            break
        beta = 0.0 if it == 0 else gamma / gamma_old
        alpha = gamma / (delta if it == 0 else delta - beta * gamma / alpha_old)
        p[own] = z[own] + beta * p[own]
        sv[own] = w + beta * sv[own]
        x[own] += alpha * p[own]
        r[own] -= alpha * sv[own]
        z0[own] = Dinv[own, own] @ r[own]
        z = precond(r, z0)
        gamma_old, alpha_old = gamma, alpha
        it += 1
    gather(x)
    return x, it


def _amg_worker(rank, world, port, out):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import dist_helpers as D
    from sim3opt_amd import lib as L
    D.init(rank, world, port)
    H, b, P = _amg_system()
    n = H.shape[0]
    lam = 1e-6 * np.abs(np.diag(H)).max()
    A = H + lam * np.eye(n)
    offs = (7 * L.partition_rows_equal(n // 7, world)).astype(np.int64)
    x, it = _amg_pcg(A, b, P, lam, int(offs[rank]), int(offs[rank + 1]), offs, rank,
                     (D.allreduce, D.allgatherv))
    if rank == 0:
        np.savez(out, x=x, it=it)


@pytest.mark.parametrize("world", [2, 8])
def test_partitioned_multigrid_pcg_over_gloo_matches_serial(tmp_path, world):
    out = str(tmp_path / "amg.npz")
    _spawn(_amg_worker, world, out)
    res = np.load(out)
    H, b, P = _amg_system()
    n = H.shape[0]
    lam = 1e-6 * np.abs(np.diag(H)).max()
    A = H + lam * np.eye(n)
    xs, its = _amg_pcg(A, b, P, lam, 0, n, None, 0, None)
    xd = np.linalg.solve(A, b)
    assert np.abs(res["x"] - xd).max() < 1e-8 * np.abs(xd).max()
    assert np.abs(xs - xd).max() < 1e-8 * np.abs(xd).max()
    assert abs(int(res["it"]) - its) <= 1  # same preconditioner, partitioned or not
    # and it is worth having: block-Jacobi PCG needs several times the iterations on this system
    Minv = np.zeros((n, n))
    for i in range(n // 7):
        Minv[7 * i:7 * i + 7, 7 * i:7 * i + 7] = np.linalg.inv(A[7 * i:7 * i + 7, 7 * i:7 * i + 7])
    x = np.zeros(n); r = b.copy(); z = Minv @ r; p = z.copy(); rz = r @ z; rz0 = rz; k = 0
    while rz > 1e-24 * rz0 and k < 5000:
        q = A @ p; a = rz / (p @ q); x += a * p; r -= a * q; z = Minv @ r
        rzn = r @ z; p = z + (rzn / rz) * p; rz = rzn; k += 1
    assert k > 3 * its
