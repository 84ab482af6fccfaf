"""CPU tests of the oracle (oracle/sim3_oracle.c) -- run with -m "not gpu".

The oracle restates g2o's LM path (see its header: parity UNPINNED against g2o itself, because
the reference cannot be built here and stores no outputs).  What pins it:
  * the reference's own input data (tests/golden/kitti00) and by-construction identities,
  * an independent numpy restatement of the Sim(3) formulae (sim3opt_amd/sim3np.py),
  * closed-form Jacobians and dense numpy linear algebra,
  * the committed vectors tests/golden/oracle_golden.json (made by tests/golden/make_golden.py).
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O
from sim3opt_amd import sim3np as S3, synth
import kitti_graph as K

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_golden.json")))


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float((np.abs(a - b) / (1.0 + np.abs(b))).max())


def rand_sim3(rng, n, rot=2.5, trans=5.0, logs=0.7):
    ax = rng.standard_normal((n, 3))
    ax /= np.linalg.norm(ax, axis=1, keepdims=True)
    xi = np.concatenate([ax * rng.uniform(0.02, rot, (n, 1)), rng.standard_normal((n, 3)) * trans,
                         rng.uniform(-logs, logs, (n, 1))], axis=1)
    return S3.exp(xi), xi


# ------------------------------------------------------------------ group arithmetic
def test_explog_golden_all_branches():
    xi = np.array(GOLD["explog"]["xi"])
    e = np.array([O.sim3_exp(x) for x in xi])
    assert rel(e, GOLD["explog"]["exp"]) < 1e-13
    assert rel(e, S3.exp(xi)) < 1e-12  # independent numpy restatement
    lg = np.array([O.sim3_log(s) for s in e])
    assert rel(lg, GOLD["explog"]["log_of_exp"]) < 1e-9
    assert rel(lg, S3.log(e)) < 1e-8


def test_exp_log_roundtrip_generic():
    rng = np.random.default_rng(1)
    S, xi = rand_sim3(rng, 200)
    back = np.array([O.sim3_log(O.sim3_exp(x)) for x in xi])
    assert np.abs(back - xi).max() < 1e-9
    S2 = np.array([O.sim3_exp(O.sim3_log(s)) for s in S])
    # q and -q are the same rotation
    sgn = np.sign((S2[:, :4] * S[:, :4]).sum(1))[:, None]
    S2[:, :4] *= sgn
    assert np.abs(S2 - S).max() < 1e-9


def test_group_identities():
    rng = np.random.default_rng(2)
    A, _ = rand_sim3(rng, 50)
    B, _ = rand_sim3(rng, 50)
    Cc, _ = rand_sim3(rng, 50)
    I = S3.identity()
    for a, b, c in zip(A, B, Cc):
        ai = O.sim3_inv(a)
        assert np.abs(O.sim3_mul(a, ai) - I).max() < 1e-12
        assert np.abs(O.sim3_mul(ai, a) - I).max() < 1e-12
        lhs = O.sim3_mul(O.sim3_mul(a, b), c)
        rhs = O.sim3_mul(a, O.sim3_mul(b, c))
        assert rel(lhs, rhs) < 1e-12
        # numpy restatement agrees
        assert rel(O.sim3_mul(a, b), S3.mul(a, b)) < 1e-14
        assert rel(ai, S3.inv(a)) < 1e-14
    # action on points: (a*b)(x) = a(b(x))
    x = rng.standard_normal(3)

    def act(s, p):
        return s[7] * (O.R_from_quat(s[:4]) @ p) + s[4:7]

    for a, b in zip(A[:10], B[:10]):
        assert np.abs(act(O.sim3_mul(a, b), x) - act(a, act(b, x))).max() < 1e-10


def test_quaternion_matrix_conversions():
    rng = np.random.default_rng(3)
    for _ in range(100):
        q = rng.standard_normal(4)
        q /= np.linalg.norm(q)
        R = O.R_from_quat(q)
        assert np.abs(R @ R.T - np.eye(3)).max() < 1e-14
        assert abs(np.linalg.det(R) - 1) < 1e-13
        q2 = O.quat_from_R(R)
        assert min(np.abs(q2 - q).max(), np.abs(q2 + q).max()) < 1e-14
    # all four branches of the matrix -> quaternion conversion (half turns about each axis)
    for ax in range(3):
        R = -np.eye(3)
        R[ax, ax] = 1
        q = O.quat_from_R(R)
        assert abs(abs(q[ax]) - 1) < 1e-15
    assert np.abs(O.euler_rpy_to_R(0.1, -0.2, 0.3) - S3.euler_rpy_to_R(0.1, -0.2, 0.3)).max() < 1e-16


def test_small_angle_b_quirk_is_restated():
    """sim3_rv.h:166 / :290 as written: B ~ 1/sigma^3 on the small-theta branch; log() reaches it
    for theta < 4.5e-3.  The oracle keeps it by default and can switch to the exact limit."""
    xi = np.array([1e-3, 0, 0, 0.3, -0.2, 0.1, 2e-2])  # theta = 1e-3 < 4.5e-3, |sigma| > 1e-5
    S = S3.exp(xi)  # exp takes the exact branch for theta >= 1e-5
    as_written = O.sim3_log(S)
    fixed = O.sim3_log(S, O.default_options(fix_small_angle_b=1))
    assert np.abs(fixed - xi).max() < 1e-6          # exact limit inverts exp
    assert np.abs(as_written[3:6] - xi[3:6]).max() > 1e-3  # as written it does not
    assert np.abs(as_written[[0, 1, 2, 6]] - xi[[0, 1, 2, 6]]).max() < 1e-6


# ------------------------------------------------------------------ edges and Jacobians
def ad_matrix(xi):
    om, up, sg = xi[:3], xi[3:6], xi[6]

    def sk(v):
        return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])

    M = np.zeros((7, 7))
    M[:3, :3] = sk(om)
    M[3:6, :3] = sk(up)
    M[3:6, 3:6] = sk(om) + sg * np.eye(3)
    M[3:6, 6] = -up
    return M


def Ad_matrix(S):
    R = S3.quat_to_R(S[:4])
    t, s = S[4:7], S[7]
    sk = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    M = np.zeros((7, 7))
    M[:3, :3] = R
    M[3:6, :3] = sk @ R
    M[3:6, 3:6] = s * R
    M[3:6, 6] = -t
    M[6, 6] = 1
    return M


def Jl(xi, terms=60):
    a = ad_matrix(xi)
    J = np.eye(7)
    T = np.eye(7)
    for n in range(1, terms):
        T = T @ a / (n + 1)
        J = J + T
    return J


def test_numeric_jacobian_matches_closed_form():
    """de/d(delta0) = Jl^-1(e) Ad_C, de/d(delta1) = -Jl^-1(-e)."""
    rng = np.random.default_rng(4)
    for _ in range(10):
        Cm, _ = rand_sim3(rng, 1, rot=1.0, trans=1.0, logs=0.2)
        S0, _ = rand_sim3(rng, 1, rot=2.0, trans=3.0, logs=0.3)
        n = np.concatenate([rng.standard_normal(3) * 0.1, rng.standard_normal(3) * 0.3,
                            rng.standard_normal(1) * 0.1])
        # S1 such that e = log(C S0 S1^-1) = n
        S1 = S3.mul(S3.inv(S3.exp(n)), S3.mul(Cm[0], S0[0]))
        e = O.edge_error(Cm[0], S0[0], S1)
        assert np.abs(e - n).max() < 1e-9
        A, B = O.edge_jacobians(Cm[0], S0[0], S1, O.default_options(fd_delta=1e-6))
        A_cf = np.linalg.solve(Jl(e), Ad_matrix(Cm[0]))
        B_cf = -np.linalg.inv(Jl(-e))
        assert np.abs(A - A_cf).max() < 1e-6 * max(1, np.abs(A_cf).max())
        assert np.abs(B - B_cf).max() < 1e-6 * max(1, np.abs(B_cf).max())
        # the g2o step 1e-9 gives the same matrix up to finite-difference noise
        A9, B9 = O.edge_jacobians(Cm[0], S0[0], S1)
        assert np.abs(A9 - A).max() < 1e-4 * max(1, np.abs(A).max())


def test_adjoint_identity():
    rng = np.random.default_rng(5)
    S, _ = rand_sim3(rng, 5)
    for s in S:
        x = rng.standard_normal(7) * 0.3
        lhs = O.sim3_mul(O.sim3_mul(s, O.sim3_exp(x)), O.sim3_inv(s))
        rhs = O.sim3_exp(Ad_matrix(s) @ x)
        sg = np.sign(lhs[:4] @ rhs[:4])
        rhs[:4] *= sg
        assert rel(lhs, rhs) < 1e-10


# ------------------------------------------------------------------ KITTI-00 pins
@pytest.mark.parametrize("one,name", [(True, "one_loop"), (False, "all_loops")])
def test_kitti_graph_pins(one, name):
    gold = GOLD["kitti"][name]
    g = K.build_direct_graph(one)
    assert g["states"].shape[0] == 771 == gold["n_vertices"]
    assert g["v0"].shape[0] == (771 if one else 888) == gold["n_edges"]
    G = O.Graph(g["states"], g["fixed"], g["v0"], g["v1"], g["meas"])
    e = G.errors()
    nl = 1 if one else 118
    # every odometry edge has zero residual by construction (kitti_surf.cpp:653-666)
    assert np.abs(e[nl:]).max() < 1e-11
    # first loop: scale ratio 5.32393351 in loopConstraints.txt record 1, frames 136 <-> 1581
    assert abs(e[0, 6] - np.log(5.32393351)) < 1e-12
    assert (g["image_ids"][g["v0"][0]], g["image_ids"][g["v1"][0]]) == (136, 1581)
    chi = G.chi2()
    assert abs(chi - gold["chi2_0"]) < 1e-9 * gold["chi2_0"]
    # values measured independently during the survey
    assert abs(chi - (169.9259622 if one else 3864464.08)) < (1e-6 if one else 1e-2)
    sel = gold["edge_sel"]
    assert np.abs(e[sel] - np.array(gold["e_sel"])).max() < 1e-11
    A, B = G.jacobians(O.default_options(fd_delta=1e-6))
    assert rel(A[sel], gold["A_sel_fd1e6"]) < 1e-7
    assert rel(B[sel], gold["B_sel_fd1e6"]) < 1e-7


def test_kitti_lm_head_matches_golden_and_survey():
    g = K.build_direct_graph(True)
    G = O.Graph(g["states"], g["fixed"], g["v0"], g["v1"], g["meas"])
    it, tr = G.optimize(4)
    assert it == 4
    chi = [t.chi2_after for t in tr]
    # first step is reproducible to ~1e-4 across implementations (28.42 in BASELINE.md);
    # later ones are dominated by finite-difference noise ("chaos" section)
    assert abs(chi[0] - 28.42) < 0.01
    assert abs(chi[0] - GOLD["kitti"]["one_loop"]["lm_chi2_head"][0]) < 1e-2
    assert all(chi[i + 1] <= chi[i] for i in range(3))


# ------------------------------------------------------------------ the LM trace, pinned independently
LMGOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kitti_lm_golden.json")))


@pytest.mark.parametrize("name,one", [("one_loop", True), ("all_loops", False)])
def test_kitti_lm_lockstep_matches_independent_lm(name, one):
    """Every LM iteration of the oracle on KITTI-00, reference configuration (delta = 1e-9, B as
    written), against ONE iteration of an independent numpy / scipy restatement started from the
    oracle's own state and lambda (tests/golden/make_lm_golden.py: numpy Sim(3) arithmetic, its own
    central differences, scipy COO assembly, SuperLU solve, its own LM policy).  An iteration ends
    with nu = 2, so (estimates, lambda) is the whole LM state and `user_lambda_init` restarts it.
    Trial counts are identical in all 20 iterations; chi2 (and with it lambda) to 1e-6 in 17 / 18 of the 20
    iterations and to 5e-4 in the sensitive ones (the configuration amplifies last-bit differences
    of libm by up to 1e3 per iteration, see the free-run test)."""
    gold = LMGOLD[name]["lockstep"]
    g = K.build_direct_graph(one)
    G = O.Graph(g["states"], g["fixed"], g["v0"], g["v1"], g["meas"])
    lam = None
    tight = 0
    for k, rec in enumerate(gold):
        o = O.default_options(user_lambda_init=lam if lam is not None else 0.0)
        it, tr = G.optimize(1, o)
        assert it == 1
        t = tr[0]
        # the fixture was generated from this very oracle: it must not have drifted
        assert abs(t.chi2_after - rec["oracle_chi2"]) <= 1e-9 * rec["oracle_chi2"], k
        assert t.trials == rec["oracle_trials"]
        # ... and the independent implementation took the same decisions from the same state
        assert t.trials == rec["trials"], (k, t.trials, rec["trials"])
        rel = abs(t.chi2_after - rec["chi2"]) / rec["chi2"]
        assert rel < 5e-4, (k, rel)
        # (lambda's factor is a function of the gain ratio, i.e. of the new chi2)
        assert abs(t.lambda_ - rec["lam"]) <= max(1e-6, 2 * rel) * rec["lam"], k
        tight += rel < 1e-6
        lam = t.lambda_
    assert tight >= 17


@pytest.mark.parametrize("name,one,same_trials", [("one_loop", True, 9), ("all_loops", False, 13)])
def test_kitti_lm_free_run_follows_independent_lm(name, one, same_trials):
    """Free runs: the oracle and the independent LM take identical accept / reject decisions for the
    first 9 (one loop) / 13 (118 loops) iterations and stay within 1e-2 in chi2 for 20; both stop the
    118-loop run through g2o's Terminate rule at iteration 40 / 41 (chi2 20.70 / 20.75; the survey's
    probe, BASELINE.md section 3, reports 18.56 after 99: a third trajectory of the same chaotic map,
    and 4.36 after iteration 2 where these two give 4.324 and 4.332)."""
    gold = LMGOLD[name]["free"]
    g = K.build_direct_graph(one)
    G = O.Graph(g["states"], g["fixed"], g["v0"], g["v1"], g["meas"])
    it, tr = G.optimize(len(gold["chi2"]))
    assert it == len(gold["chi2"])
    assert [t.trials for t in tr[:same_trials]] == gold["trials"][:same_trials]
    for k in range(it):
        assert abs(tr[k].chi2_after - gold["chi2"][k]) < 1e-2 * gold["chi2"][k], k
    for k in range(same_trials):
        assert abs(tr[k].lambda_ - gold["lam"][k]) < 1e-2 * gold["lam"][k], k
    # final poses of the two 20-iteration runs: the CPU-vs-CPU spread of this configuration
    t_wi = S3.inv(G.states)[:, 4:7]
    rm = np.sqrt(((t_wi - np.array(gold["t_wi"])) ** 2).sum(1).mean())
    assert rm < (2e-3 if one else 0.5), rm


# ------------------------------------------------------------------ normal equations, solver, LM
def small_graph(seed=0, info=False, kernel=0):
    synth.DRIFT_TARGET = 0.05
    g = synth.manhattan(40, 200, dims=(4, 4, 2), per_cell=4, seed_graph=100 + seed,
                        seed_noise=200 + seed)
    rng = np.random.default_rng(seed)
    inf = None
    if info:
        M = rng.standard_normal((g["v0"].shape[0], 7, 7)) * 0.3
        inf = np.einsum("kij,klj->kil", M, M) + np.eye(7)
        inf = inf.transpose(0, 2, 1).reshape(-1, 49)  # column-major blocks
    return g, O.Graph(g["states"], g["fixed"], g["v0"], g["v1"], g["meas"], info=inf,
                      kernel=kernel, kdelta=0.08 if kernel else 0.0)


@pytest.mark.parametrize("info,kernel", [(False, 0), (True, 0), (False, 1), (True, 1)])
def test_dense_system_matches_numpy_assembly(info, kernel):
    g, G = small_graph(1, info, kernel)
    H, b = G.build_dense()
    e = G.errors()
    A, B = G.jacobians()
    free = np.cumsum(1 - g["fixed"].astype(int)) - 1
    free[g["fixed"] == 1] = -1
    n = 7 * G.n_free
    H2 = np.zeros((n, n))
    b2 = np.zeros(n)
    chi_sum = 0.0
    for k in range(G.ne):
        Om = np.eye(7) if G.info is None else G.info[k].reshape(7, 7).T
        chi = e[k] @ Om @ e[k]
        w = 1.0
        rho = chi
        if kernel and chi > G.kdelta ** 2:
            w = G.kdelta / np.sqrt(chi)
            rho = 2 * np.sqrt(chi) * G.kdelta - G.kdelta ** 2
        chi_sum += rho
        i, j = free[g["v0"][k]], free[g["v1"][k]]
        if i >= 0:
            H2[7 * i:7 * i + 7, 7 * i:7 * i + 7] += w * A[k].T @ Om @ A[k]
            b2[7 * i:7 * i + 7] -= w * A[k].T @ Om @ e[k]
        if j >= 0:
            H2[7 * j:7 * j + 7, 7 * j:7 * j + 7] += w * B[k].T @ Om @ B[k]
            b2[7 * j:7 * j + 7] -= w * B[k].T @ Om @ e[k]
        if i >= 0 and j >= 0:
            H2[7 * i:7 * i + 7, 7 * j:7 * j + 7] += w * A[k].T @ Om @ B[k]
            H2[7 * j:7 * j + 7, 7 * i:7 * i + 7] += w * B[k].T @ Om @ A[k]
    assert np.abs(H - H2).max() < 1e-9 * np.abs(H2).max()
    assert np.abs(b - b2).max() < 1e-9 * max(1, np.abs(b2).max())
    assert np.abs(H - H.T).max() < 1e-9 * np.abs(H).max()
    assert abs(G.chi2() - chi_sum) < 1e-10 * max(1, chi_sum)


def test_sparse_ldlt_matches_dense_solve():
    for seed in range(3):
        g, G = small_graph(seed)
        H, b = G.build_dense()
        lam = 1e-5 * np.abs(np.diag(H)).max()
        ok, x, b2 = G.solve_once(lam)
        assert ok
        assert np.abs(b - b2).max() == 0
        xd = np.linalg.solve(H + lam * np.eye(H.shape[0]), b)
        assert np.abs(x - xd).max() < 1e-8 * np.abs(xd).max()
    g = K.build_direct_graph(False)
    G = O.Graph(g["states"], g["fixed"], g["v0"], g["v1"], g["meas"])
    H, b = G.build_dense()
    lam = 1e-5 * np.abs(np.diag(H)).max()
    ok, x, _ = G.solve_once(lam)
    xd = np.linalg.solve(H + lam * np.eye(H.shape[0]), b)
    assert ok and np.abs(x - xd).max() < 1e-6 * np.abs(xd).max()


def test_ldlt_rejects_indefinite_system():
    g, G = small_graph(0)
    ok, x, _ = G.solve_once(-1e9)
    assert not ok


def test_lm_policy_and_return_codes():
    g, G = small_graph(2)
    it, tr = G.optimize(6, O.default_options(fix_small_angle_b=1))
    assert it == 6
    H, _ = G0 = O.Graph(g["states"], g["fixed"], g["v0"], g["v1"], g["meas"]).build_dense(
        O.default_options(fix_small_angle_b=1))
    # lambda_0 = tau * max diag, then one accept/reject update
    lam0 = 1e-5 * np.abs(np.diag(H)).max()
    assert tr[0].trials >= 1 and tr[0].lambda_ > 0
    if tr[0].trials == 1:
        assert 1 / 3 - 1e-12 <= tr[0].lambda_ / lam0 <= 2 / 3 + 1e-12
    assert all(t.chi2_after <= t.chi2_before + 1e-12 for t in tr)
    # nothing to optimise -> -1 (g2o convention)
    Gf = O.Graph(g["states"], np.ones_like(g["fixed"]), g["v0"], g["v1"], g["meas"])
    assert Gf.optimize(3)[0] == -1


@pytest.mark.parametrize("name", ["manhattan_120", "chain_150"])
def test_synthetic_golden_fixb(name):
    gold = GOLD["synthetic_fixb"][name]["fd1e9"]
    synth.DRIFT_TARGET = 0.05
    g = (synth.manhattan(120, 1000, dims=(6, 6, 3), per_cell=4) if name == "manhattan_120"
         else synth.chain_loop(150, 300))
    G = O.Graph(g["states"], g["fixed"], g["v0"], g["v1"], g["meas"])
    o = O.default_options(fix_small_angle_b=1)
    assert abs(G.chi2(o) - gold["chi2_0"]) < 1e-9 * gold["chi2_0"]
    it, tr = G.optimize(15, o)
    assert it == gold["iters"]
    assert abs(tr[-1].chi2_after - gold["chi2_final"]) < 1e-6 * gold["chi2_final"]
    d = synth.positions(G.states) - np.array(gold["positions"])
    assert np.sqrt((d ** 2).sum(1).mean()) < 1e-6
    # the optimum is a genuine improvement over the noisy initial guess
    assert tr[-1].chi2_after < 0.5 * gold["chi2_0"]


def test_openmp_variant_is_identical():
    g, G1 = small_graph(3)
    _, G8 = small_graph(3)
    a = G1.optimize(3, O.default_options(threads=1))
    b = G8.optimize(3, O.default_options(threads=4))
    assert np.array_equal(G1.states, G8.states)


def test_dof_mask_freezes_tangent_components():
    """Stage 2 of the stepwise pipeline: cleared dof_mask bits zero the Jacobian columns."""
    g, G = small_graph(4)
    o = O.default_options(dof_mask=0x78, fix_small_angle_b=1)
    A, B = G.jacobians(o)
    assert np.all(A[:, :, :3] == 0) and np.all(B[:, :, :3] == 0)
    assert np.abs(A[:, :, 3:]).max() > 0
    q0 = G.states[:, :4].copy()
    it, tr = G.optimize(4, o)
    assert it == 4 and np.array_equal(G.states[:, :4], q0)
    assert tr[-1].chi2_after < tr[0].chi2_before
