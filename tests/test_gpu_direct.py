"""GPU tests (-m gpu) of the exact sparse block Cholesky (LinearSolverEigen's role,
kitti_surf.cpp:553-554) and of KITTI-00 in the reference's OWN configuration: delta = 1e-9 numeric
Jacobians, the small-angle B coefficient as written (sim3_rv.h:166, :290), optimize(100)
(kitti_surf.cpp:674-675), first loop only (:1317) and all 118 loops.

What "parity" can mean there: the configuration amplifies last-bit differences
of libm by up to 1e3 per LM iteration, so
  * LOCK-STEP (the rigorous check): every one of the first 20 iterations restarted from the
    oracle's state and lambda -- identical trial counts, lambda and chi2 as close as the independent
    numpy / scipy LM of tests/golden/make_lm_golden.py gets (tests/test_oracle.py);
  * FREE RUN: the device evaluates residuals and numeric Jacobians in the oracle's operation order
    without FMA contraction (sim3_math.hpp), so H and b agree to 1e-13 and the free runs stay
    together: one loop -- 100 iterations on both sides, identical accept / reject decisions for 15,
    final chi2 to 1.4e-6, trajectory RMSE 5.9e-6 m (north_star: < 1e-4); 118 loops -- identical
    decisions for 11 iterations, Terminate at 39 / 41, RMSE 7.7e-4 m (the oracle's own sensitivity
    there is 1.5 m)."""
import json
import os
import time

import numpy as np
import pytest

from oracle import oracle as O
from sim3opt_amd import lib as L, sim3np as S3, synth
import kitti_graph as K

pytestmark = pytest.mark.gpu
LMGOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kitti_lm_golden.json")))


def mk(g, **opts):
    G = L.Graph(**opts)
    G.add_vertices(g["states"], g["fixed"])
    G.add_edges(g["v0"], g["v1"], g["meas"])
    G.initialize()
    return G


def oracle_of(g):
    return O.Graph(g["states"], g["fixed"], g["v0"], g["v1"], g["meas"])


# ------------------------------------------------------------------ the factorisation itself
CASES = {
    "kitti_one_loop": (lambda: K.build_direct_graph(True), {}),
    "kitti_all_loops": (lambda: K.build_direct_graph(False), {}),
    "chain_200": (lambda: synth.chain_loop(200, 230), {}),
    "tiny_5": (lambda: synth.chain_loop(5, 6, min_gap=2), {}),
    # heavy fill: only on request
    "manhattan_300": (lambda: synth.manhattan(300, 1500, dims=(8, 8, 3)), dict(linear_solver=1)),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_direct_solve_matches_dense(name):
    make, opts = CASES[name]
    G = mk(make(), fix_small_angle_b=1, **opts)
    assert G.linear_solver_in_use() == 1
    G.linearize()
    H, b = G.dense_system()
    n = H.shape[0]
    for lam in (1e-2, 1.0, 1e3):
        x, it, _ = G.solve(lam)
        A = H + lam * np.eye(n)
        xr = np.linalg.solve(A, b)
        assert it == 0  # no iterations: a factorisation
        assert np.abs(A @ x - b).max() < 1e-12 * max(np.abs(b).max(), 1e-300) * n
        assert np.abs(x - xr).max() < 1e-8 * np.abs(xr).max()
    x2, _, _ = G.solve(1e3)
    assert np.array_equal(x, x2)  # fixed summation order: bit-reproducible


def test_direct_results_do_not_depend_on_the_schedule(monkeypatch):
    """The plan's knobs (bottom-subtree size, wavefronts per bottom group) change which workgroup
    eliminates what and when, never the order in which a block's products or a column's backward
    terms are summed (direct.cpp: the reference numbering): the step is the same BITS for every
    schedule, so tuning the schedule cannot move a chaotic LM run (KITTI-00, delta = 1e-9)."""
    g = K.build_direct_graph(False)  # all 118 loops: the widest separators
    ref = None
    groups = set()
    for subtree, wg in (("16", "256"), ("48", "512"), ("128", "384"), ("8", "64")):
        monkeypatch.setenv("SIM3OPT_DIRECT_SUBTREE", subtree)
        monkeypatch.setenv("SIM3OPT_DIRECT_WG_SUB", wg)
        G = mk(g, linear_solver=1)
        groups.add(G.direct_plan()["ngroups"])  # (the host plan reads the same knob)
        G.linearize()
        x = G.solve(1e-3)[0].copy()
        n = G.optimize(6)
        st = [(s.trials, s.chi2_after, s.lambda_) for s in G.stats()]
        V = np.array(G.get_vertices(), copy=True)
        G.close()
        if ref is None:
            ref = (x, n, st, V)
            continue
        assert np.array_equal(x, ref[0])  # bitwise
        assert n == ref[1] and st == ref[2] and np.array_equal(V, ref[3])
    assert len(groups) > 1


def test_direct_reports_indefinite_system():
    """Non-positive pivot = g2o's `solve` returning false: the C-ABI reports it, LM would reject."""
    G = mk(K.build_direct_graph(True))
    assert G.linear_solver_in_use() == 1
    G.linearize()
    with pytest.raises(L.Sim3OptError):
        G.solve(-1e15)
    x, _, _ = G.solve(1.0)  # the next solve is unaffected
    assert np.isfinite(x).all()


def test_linear_solver_option():
    g = synth.chain_loop(200, 230)
    assert mk(g, linear_solver=0).linear_solver_in_use() == 0
    assert mk(g, linear_solver=-1).linear_solver_in_use() == 1
    # heavy fill: automatic = PCG, forced = exact
    g = synth.manhattan(1000, 10000, dims=(10, 10, 5))
    assert mk(g).linear_solver_in_use() == 0
    # exact and iterative steps agree on a well-posed graph
    synth.DRIFT_TARGET = 0.05
    g = synth.manhattan(150, 900, dims=(5, 5, 3), per_cell=4)
    Gd = mk(g, linear_solver=1, fix_small_angle_b=1, fd_delta=1e-6)
    Gi = mk(g, linear_solver=0, fix_small_angle_b=1, fd_delta=1e-6, pcg_rel_tol=1e-13)
    assert Gd.optimize(6) == Gi.optimize(6) == 6
    assert abs(Gd.stats()[-1].chi2_after - Gi.stats()[-1].chi2_after) < 1e-8 * Gi.stats()[-1].chi2_after
    assert synth.rmse(Gd.get_vertices(), Gi.get_vertices()) < 1e-6


def test_direct_lm_with_information_huber_parallel_edges_and_two_fixed_vertices():
    """The exact solver under everything the edge / vertex surface offers at once: non-identity
    information matrices, a Huber kernel, duplicated (parallel) edges, two fixed vertices (one in the
    middle of the chain), arbitrary vertex ids -- LM against the oracle in the well-posed arithmetic."""
    synth.DRIFT_TARGET = 0.05
    g = synth.chain_loop(300, 340)
    rng = np.random.default_rng(5)
    dup = rng.choice(g["v0"].shape[0], 25, replace=False)  # parallel edges
    v0 = np.concatenate([g["v0"], g["v0"][dup]]).astype(np.int32)
    v1 = np.concatenate([g["v1"], g["v1"][dup]]).astype(np.int32)
    meas = np.concatenate([g["meas"], g["meas"][dup]])
    fixed = g["fixed"].copy()
    fixed[150] = 1
    M = rng.standard_normal((v0.shape[0], 7, 7)) * 0.3
    info = np.einsum("kij,klj->kil", M, M) + np.eye(7)
    ids = (np.arange(300) * 7 + 3).astype(np.int32)
    G = L.Graph(fix_small_angle_b=1, fd_delta=1e-6)
    G.add_vertices(g["states"], fixed, ids)
    G.add_edges(ids[v0], ids[v1], meas, info=info, kernel=L.KERNEL_HUBER, kernel_delta=0.3)
    G.initialize()
    assert G.linear_solver_in_use() == 1
    OG = O.Graph(g["states"], fixed, v0, v1, meas, info=info.transpose(0, 2, 1).reshape(-1, 49),
                 kernel=1, kdelta=0.3)
    o = O.default_options(fix_small_angle_b=1, fd_delta=1e-6)
    n = G.optimize(8)
    it, tr = OG.optimize(8, o)
    st = G.stats()
    assert n == it == 8
    assert [s.trials for s in st] == [t.trials for t in tr]
    for k in range(8):
        assert abs(st[k].chi2_after - tr[k].chi2_after) < 1e-7 * tr[k].chi2_after, k
    assert synth.rmse(G.get_vertices(), OG.states) < 1e-6
    assert np.array_equal(G.get_vertices()[150], g["states"][150])  # the fixed vertex did not move


# ------------------------------------------------------------------ KITTI-00, reference configuration
@pytest.mark.parametrize("name,one", [("one_loop", True), ("all_loops", False)])
def test_kitti_lockstep_with_oracle(name, one):
    """Each of the first 20 LM iterations from the oracle's state and lambda: same trial counts as the
    oracle (and as the independent LM of the fixture), chi2 to 1e-6 in most iterations and to 2e-3
    in the sensitive ones.  There the device's H and b equal the oracle's to 1e-13 (the Sim(3)
    arithmetic is restated operation by operation, scripts/gpu_bitparity.py), and the remaining
    difference is the rounding of two exact factorisations of a matrix with cond ~ 4e11 (different
    elimination orders): the independent SuperLU solve of the fixture differs from the oracle's
    LDL^T by as much."""
    gold = LMGOLD[name]["lockstep"]
    g = K.build_direct_graph(one)
    OG = oracle_of(g)
    G = mk(g)
    assert G.linear_solver_in_use() == 1
    lam = None
    tight = 0
    worst = 0.0
    for k, rec in enumerate(gold):
        G.set_vertices(OG.states)
        G.set_options(user_lambda_init=lam if lam is not None else 0.0)
        assert G.optimize(1) == 1
        s = G.stats()[0]
        it, tr = OG.optimize(1, O.default_options(user_lambda_init=lam if lam is not None else 0.0))
        t = tr[0]
        assert t.trials == rec["oracle_trials"]  # the oracle is where the fixture left it
        assert s.trials == t.trials, (k, s.trials, t.trials)
        rel = abs(s.chi2_after - t.chi2_after) / t.chi2_after
        worst = max(worst, rel)
        assert rel < 2e-3, (k, rel)
        assert abs(s.lambda_ - t.lambda_) <= max(1e-6, 2 * rel) * t.lambda_, k
        tight += rel < 1e-6
        lam = t.lambda_
    assert tight >= 15, (tight, worst)


def test_kitti_one_loop_reference_run():
    """optimize(100) as kitti_surf.cpp:675 calls it, default options, `bUseOneContraint` graph
    (BASELINE.json configs[0] the way case 3 runs it, kitti_surf.cpp:1317): the north_star bar --
    trajectory RMSE against the reference solver below 1e-4 m (measured 5.9e-6), same number of
    iterations, identical accept / reject decisions for the first 12 iterations (measured: 15), final
    chi2 to 1e-4 (measured 1.4e-6).  Wall-clock times are printed, not asserted (a busy host must not
    turn a performance wobble into a parity failure; bench.py reports the ratio: 10x)."""
    g = K.build_direct_graph(True)
    OG = oracle_of(g)
    t0 = time.perf_counter()
    it, tr = OG.optimize(100)
    t_cpu = time.perf_counter() - t0
    G = mk(g)
    G.optimize(2)  # warm-up (first launches, clocks)
    G.set_vertices(g["states"])
    t0 = time.perf_counter()
    n = G.optimize(100)
    t_gpu = time.perf_counter() - t0
    st = G.stats()
    assert n == it == 100
    assert [s.trials for s in st[:12]] == [t.trials for t in tr[:12]]
    assert [s.trials for s in st[:9]] == [1, 1, 5, 1, 2, 7, 1, 1, 3]
    for k in range(12):
        assert abs(st[k].chi2_after - tr[k].chi2_after) < 1e-3 * tr[k].chi2_after, k
    assert abs(st[-1].chi2_after - tr[-1].chi2_after) < 1e-4 * tr[-1].chi2_after
    assert abs(st[-1].chi2_after - 1.2091) < 1e-3
    rm = synth.rmse(G.get_vertices(), OG.states)
    assert rm < 1e-4, rm
    print(f"KITTI-00 one loop, optimize(100): GPU {t_gpu:.3f} s, oracle on one CPU thread {t_cpu:.3f} s")
    # a second run from the same start is bit-identical
    G.set_vertices(g["states"])
    assert G.optimize(100) == 100
    assert [s.chi2_after for s in G.stats()] == [s.chi2_after for s in st]


def test_kitti_all_loops_reference_run():
    """All 118 loop constraints (`bUseOneContraint = false`).  The run is more sensitive than the
    one-loop one (the oracle itself moves by 1.5 m RMSE under a 1e-15 input perturbation, and the
    independent numpy LM of the fixture ends at chi2 20.70 where the oracle ends at 20.75): measured
    here 7.7e-4 m RMSE against the oracle, identical decisions for 11 iterations, Terminate at 39
    against 41, final chi2 within 1.7e-3.  The bounds are ~3x what is measured; this configuration is
    never quoted as "parity" (DESIGN.md section 2)."""
    g = K.build_direct_graph(False)
    OG = oracle_of(g)
    it, tr = OG.optimize(100)
    G = mk(g)
    n = G.optimize(100)
    st = G.stats()
    for k in range(5):
        assert abs(st[k].chi2_after - tr[k].chi2_after) < 1e-4 * tr[k].chi2_after, k
    for k in range(10):
        assert abs(st[k].chi2_after - tr[k].chi2_after) < 2e-2 * tr[k].chi2_after, k
    assert [s.trials for s in st[:10]] == [t.trials for t in tr[:10]]
    assert n < 100 and it < 100 and abs(n - it) <= 5  # g2o's Terminate rule fires on both sides
    assert abs(st[-1].chi2_after - tr[-1].chi2_after) < 5e-3 * tr[-1].chi2_after
    rm = synth.rmse(G.get_vertices(), OG.states)
    assert rm < 2.5e-3, rm


@pytest.mark.parametrize("one,rmse_max,chi_rel", [(True, 1e-4, 1e-6), (False, 6e-2, 2e-4)])
def test_kitti_stepwise_reference_run(one, rmse_max, chi_rel):
    """BASELINE.json configs[4] in the reference's meaning (kitti_surf.cpp:887-1047): scales from the
    null vector, scale + translation LM with frozen rotations (100 it), Sim(3) LM warm-started from it
    (100 it), default options (delta = 1e-9, B as written), through the exact Cholesky.  Measured
    against the oracle from the same scale initialisation: RMSE 3.9e-5 m (one loop) / 0.02 m (118
    loops), final chi2 to 2e-8 / 5e-5; against KITTI ground truth both land at 13.4 m with all loops
    (the direct run: 117 m) -- the staging is what makes the reference's result usable."""
    g = K.build_direct_graph(one)
    G = L.Graph()
    G.add_vertices(g["states"], g["fixed"])
    G.add_edges(g["v0"], g["v1"], g["meas"])
    G.stepwise_scale_init()
    st0 = G.get_vertices().copy()
    G.set_options(dof_mask=0x78)
    G.initialize()
    assert G.linear_solver_in_use() == 1
    q0 = G.get_vertices()[:, :4].copy()
    assert G.optimize(100) > 0
    assert np.array_equal(G.get_vertices()[:, :4], q0)  # rotations frozen
    chi_st = G.stats()[-1].chi2_after
    G.set_options(dof_mask=127)
    assert G.optimize(100) > 0
    OG = O.Graph(st0, g["fixed"], g["v0"], g["v1"], g["meas"])
    i2, tr2 = OG.optimize(100, O.default_options(dof_mask=0x78))
    i3, tr3 = OG.optimize(100)
    assert abs(chi_st - tr2[-1].chi2_after) < chi_rel * tr2[-1].chi2_after
    assert abs(G.stats()[-1].chi2_after - tr3[-1].chi2_after) < chi_rel * tr3[-1].chi2_after
    rm = synth.rmse(G.get_vertices(), OG.states)
    assert rm < rmse_max, rm
    if not one:
        gt = np.loadtxt(os.path.join(K.FIXTURE, "gt_kf.txt"), comments="%")[:, [4, 8, 12]]
        assert L.align_trajectory(synth.positions(G.get_vertices()), gt)[1] < 20.0


def test_kitti_incremental_closures_reference_arithmetic():
    """BASELINE.json configs[4] in its own words, reference configuration: the first closures of
    loopConstraints.txt added one at a time, optimize(100) warm-started after each.  What g2o's rules
    do with the as-written small-angle coefficient: the first closure runs its 100 iterations, every
    later one ends after a single iteration (ten rejected trials) -- on the GPU exactly as in the
    oracle (scripts/gpu_incremental.py runs all 118 and both arithmetics)."""
    full = K.build_direct_graph(False)
    nl = 118
    G = L.Graph()
    G.add_vertices(full["states"], full["fixed"])
    G.add_edges(full["v0"][nl:], full["v1"][nl:], full["meas"][nl:])
    states = full["states"].copy()
    gi, ci = [], []
    for k in range(4):
        G.add_edge(int(full["v0"][k]), int(full["v1"][k]), full["meas"][k])
        G.initialize()
        assert G.linear_solver_in_use() == 1
        gi.append(G.optimize(100))
        idx = np.r_[np.arange(nl, len(full["v0"])), np.arange(k + 1)]
        OG = O.Graph(states, full["fixed"], full["v0"][idx], full["v1"][idx], full["meas"][idx])
        it, tr = OG.optimize(100)
        states = OG.states.copy()
        ci.append(it)
        assert abs(G.stats()[-1].chi2_after - tr[-1].chi2_after) < 2e-3 * tr[-1].chi2_after
        assert G.stats()[-1].trials == tr[-1].trials
    assert gi == ci == [100, 1, 1, 1]
    assert synth.rmse(G.get_vertices(), states) < 1e-3


@pytest.mark.parametrize("fixb", [0, 1])
def test_kitti_incremental_all_118_closures_lockstep(fixb):
    """BASELINE.json configs[4] in its own words, WHOLE: all 118 closures of loopConstraints.txt added one
    at a time, optimize(100) after each (kitti_surf.cpp:1028-1047 is the warm start being matched), in the
    reference's arithmetic (fixb = 0: B as written) and in the exact one.  Lock-step: every closure starts
    both sides from the ORACLE's previous solution, so each of the 118 runs is compared on its own -- a
    free run of 118 x 100 chaotic iterations drifts apart after the eleventh closure (DESIGN.md 2;
    scripts/gpu_incremental.py runs it free).  Asserted per closure: LM iteration count, trials of the
    last iteration, final chi2."""
    full = K.build_direct_graph(False)
    nl = 118
    G = L.Graph(fix_small_angle_b=fixb)
    G.add_vertices(full["states"], full["fixed"])
    G.add_edges(full["v0"][nl:], full["v1"][nl:], full["meas"][nl:])
    o = O.default_options(fix_small_angle_b=fixb)
    states = full["states"].copy()
    rec = []
    for k in range(nl):
        G.add_edge(int(full["v0"][k]), int(full["v1"][k]), full["meas"][k])
        G.initialize()
        G.set_vertices(states)
        assert G.linear_solver_in_use() == 1
        n = G.optimize(100)
        st = G.stats()
        idx = np.r_[np.arange(nl, len(full["v0"])), np.arange(k + 1)]
        OG = O.Graph(states, full["fixed"], full["v0"][idx], full["v1"][idx], full["meas"][idx])
        it, tr = OG.optimize(100, o)
        rec.append((n, it, st[-1].trials, tr[-1].trials, st[-1].chi2_after, tr[-1].chi2_after,
                    synth.rmse(G.get_vertices(), OG.states)))
        states = OG.states.copy()
    rec = np.array(rec)
    same_it = int((rec[:, 0] == rec[:, 1]).sum())
    same_tr = int((rec[:, 2] == rec[:, 3]).sum())
    rel = np.abs(rec[:, 4] - rec[:, 5]) / np.maximum(rec[:, 5], 1e-12)
    print(f"fix_small_angle_b={fixb}: closures with equal LM iteration count {same_it}/118, equal last-iteration "
          f"trials {same_tr}/118, LM iterations GPU/oracle {int(rec[:, 0].sum())}/{int(rec[:, 1].sum())}, "
          f"chi2 rel. diff median {np.median(rel):.1e} max {rel.max():.1e}, RMSE GPU vs oracle per closure "
          f"median {np.median(rec[:, 6]):.1e} max {rec[:, 6].max():.1e}, unequal at {np.nonzero(rec[:, 0] != rec[:, 1])[0].tolist()}")
    bounds = INCREMENTAL_BOUNDS[fixb]
    assert same_it >= bounds["same_it"] and same_tr >= bounds["same_tr"]
    assert np.median(rel) < bounds["chi_med"] and rel.max() < bounds["chi_max"]
    assert np.median(rec[:, 6]) < bounds["rmse_med"]


# per arithmetic: what one MI355X run measured (profiles/r4_incremental_lockstep.log), bounds ~3x of it
# measured: reference arithmetic -- 112 / 118 closures with the oracle's LM iteration count (most end after ONE
# iteration of ten rejected trials), 107 with its last-iteration trial count, chi2 relative
# difference median 1.5e-16 (max 7.5e-2 where a 100-iteration episode starts on one side only), RMSE median 0;
# exact B -- every closure converges, the Terminate iteration of a converged run is decided at the noise floor
# of the delta = 1e-9 Jacobians (equal in 15 closures; 5684 against 5582 LM iterations in all), chi2 relative
# difference median 2.3e-6 (max 1.3e-3), RMSE per closure median 5e-3 m
INCREMENTAL_BOUNDS = {
    0: dict(same_it=105, same_tr=98, chi_med=1e-12, chi_max=0.25, rmse_med=1e-9),
    1: dict(same_it=6, same_tr=8, chi_med=1e-5, chi_max=5e-3, rmse_med=2e-2),
}
