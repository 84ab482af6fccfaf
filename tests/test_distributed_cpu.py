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
    D.finish()


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
# the multigrid-preconditioned PCG, row-partitioned (Engine::amg_apply, multi-GPU branch; round 4): block rows in
# locality order, aggregates inside the ranks' spans, so that a rank forms the Galerkin rows and the restricted
# residual of ITS coarse rows completely -- no reduction across ranks; the replicated coarse level receives the
# owners' pieces by all-gather; level-0 iterates travel by the neighbour exchange of the library's halo plan
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


def _amg_system(world):
    from oracle import oracle as O
    from sim3opt_amd import lib as L, synth
    synth.DRIFT_TARGET = 0.05
    g = synth.manhattan(400, 4000, dims=(6, 6, 10))
    H, b = O.Graph(g["states"], g["fixed"], g["v0"], g["v1"], g["meas"]).build_dense(
        O.default_options(fix_small_angle_b=1, fd_delta=1e-6))
    # the library's own row order, partition, aggregation and halo plan (host code, no GPU)
    G = L.Graph(row_order=1, amg_virtual_ranks=world)
    G.add_vertices(g["states"], g["fixed"])
    G.add_edges(g["v0"], g["v1"], g["meas"])
    rows, _, agg = G.amg_hierarchy()
    nb = rows[0]
    assert len(rows) == 2 and rows[1] <= 256
    vor, beg, _, _ = G.partition_plan(world)          # vertex of every block row (locality order), rank spans
    free = np.where(g["fixed"] == 0)[0]
    pos = {int(v): k for k, v in enumerate(free)}      # build_dense numbers the free vertices in insertion order
    perm = np.array([pos[int(v)] for v in vor])
    idx = (7 * perm[:, None] + np.arange(7)[None, :]).reshape(-1)
    H, b = H[np.ix_(idx, idx)], b[idx]
    Ad = _adjoint(g["states"][vor])
    P = np.zeros((7 * nb, 7 * rows[1]))
    for i in range(nb):
        P[7 * i:7 * i + 7, 7 * agg[i]:7 * agg[i] + 7] = Ad[i]
    # coarse rows of every rank: the aggregates of its own rows -- contiguous, and no aggregate straddles
    cbeg = [0]
    for r in range(world):
        a = np.unique(agg[beg[r]:beg[r + 1]])
        assert a.size == 0 or (a[0] == cbeg[-1] and a[-1] - a[0] + 1 == a.size)
        cbeg.append(cbeg[-1] + a.size)
    assert cbeg[-1] == rows[1]
    plans = [G.halo_plan(world, r) for r in range(world)]
    return H, b, P, np.asarray(beg), np.asarray(cbeg), plans


def _amg_pcg(A, b, P, lam, beg, cbeg, plan, rank, coll):
    """Single-reduction PCG with the two-level multiplicative cycle; block rows [beg[rank], beg[rank+1]) are this
    rank's.  coll = (allreduce, allgatherv, alltoallv) or None for a serial run."""
    n, nc = A.shape[0], P.shape[1]
    omega = 0.9
    lo, hi = (7 * int(beg[rank]), 7 * int(beg[rank + 1])) if coll else (0, n)
    clo, chi = (7 * int(cbeg[rank]), 7 * int(cbeg[rank + 1])) if coll else (0, nc)
    own = slice(lo, hi)
    Dinv = np.zeros((n, n))
    for i in range(lo // 7, hi // 7):
        Dinv[7 * i:7 * i + 7, 7 * i:7 * i + 7] = omega * np.linalg.inv(A[7 * i:7 * i + 7, 7 * i:7 * i + 7])
    # Galerkin product of this rank's rows: COMPLETE rows of its own aggregates, nothing elsewhere
    Ac = P[own].T @ (A[own] - lam * np.eye(n)[own]) @ P
    if coll:
        outside = np.ones(nc, bool)
        outside[clo:chi] = False
        assert not Ac[outside].any()
        coll[1](Ac.reshape(-1), (nc * 7 * np.asarray(cbeg)).astype(np.int64), rank)   # owners' rows, all-gathered
    Ainv_c = np.linalg.inv(Ac + lam * (P.T @ P))  # damping carried as lambda W, W = P^T P (replicated)
    if coll:
        srows, sseg, rrows, rseg = plan
        sidx = (7 * srows[:, None] + np.arange(7)[None, :]).reshape(-1)
        ridx = (7 * rrows[:, None] + np.arange(7)[None, :]).reshape(-1)
        soffs, roffs = (7 * sseg).astype(np.int64), (7 * rseg).astype(np.int64)
        need = np.zeros(n, bool)  # the rows this rank's blocks read
        need[lo:hi] = True
        need[ridx] = True
        assert not (A[own][:, ~need] != 0).any()

    def exchange(v):  # neighbour exchange: exactly the rows the other side reads
        if coll:
            recv = np.zeros(max(ridx.size, 1))
            coll[2](np.ascontiguousarray(v[sidx]) if sidx.size else np.zeros(1), soffs, recv, roffs, rank)
            v[ridx] = recv[:ridx.size]

    def precond(r, z0):  # z0 = omega D^-1 r on own rows (what the PCG step hands over)
        x = z0.copy()
        exchange(x)
        t = r[own] - A[own] @ x
        r1 = P[own].T @ t              # complete on this rank's coarse rows, zero elsewhere
        if coll:
            coll[1](r1, (7 * np.asarray(cbeg)).astype(np.int64), rank)
        xc = Ainv_c @ r1               # replicated coarse level
        x = x + P @ xc                 # own rows and the foreign rows read: xc is replicated, no collective
        out = np.zeros(n)
        out[own] = x[own] + Dinv[own, own] @ (r[own] - A[own] @ x)
        exchange(out)
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
    if coll:
        coll[1](x, (7 * np.asarray(beg)).astype(np.int64), rank)
    return x, it


def _amg_worker(rank, world, port, out):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import dist_helpers as D
    D.init(rank, world, port)
    H, b, P, beg, cbeg, plans = _amg_system(world)
    n = H.shape[0]
    lam = 1e-6 * np.abs(np.diag(H)).max()
    A = H + lam * np.eye(n)
    x, it = _amg_pcg(A, b, P, lam, beg, cbeg, plans[rank], rank, (D.allreduce, D.allgatherv, D.alltoallv))
    if rank == 0:
        np.savez(out, x=x, it=it)
    D.finish()


@pytest.mark.parametrize("world", [2, 8])
def test_partitioned_multigrid_pcg_over_gloo_matches_serial(tmp_path, world):
    out = str(tmp_path / "amg.npz")
    _spawn(_amg_worker, world, out)
    res = np.load(out)
    H, b, P, beg, cbeg, plans = _amg_system(world)
    n = H.shape[0]
    lam = 1e-6 * np.abs(np.diag(H)).max()
    A = H + lam * np.eye(n)
    # the halo plans of two ranks mirror each other: what p receives from q is what q sends to p, in order
    for p in range(world):
        for q in range(world):
            sp, ssp, rp, rsp = plans[p]
            sq, ssq, rq, rsq = plans[q]
            assert np.array_equal(rp[rsp[q]:rsp[q + 1]], sq[ssq[p]:ssq[p + 1]])
            assert p != q or rsp[q + 1] == rsp[q]
    xs, its = _amg_pcg(A, b, P, lam, beg, cbeg, None, 0, None)
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
