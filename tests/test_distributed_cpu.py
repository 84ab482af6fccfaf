"""world_size-2 gloo test of the row-partitioned PCG (CPU).

The HIP kernels cannot run here, so this test executes the distributed ALGORITHM the engine uses
(Engine::pcg in csrc/engine.hip, multi-GPU branch) in numpy, with the engine's own partition
function (sim3opt_partition_rows_equal through the C-ABI) and the same collectives in the same order --
all-gather of z and one 2-double all-reduce of (w.z, r.z) per iteration -- over torch.distributed gloo,
and checks it against a serial solve.  The real kernels run the same path with 2 processes on one
GPU in tests/test_distributed_gpu.py.
"""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


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


def test_partitioned_pcg_over_gloo_matches_serial(tmp_path):
    out = str(tmp_path / "x.npy")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
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
