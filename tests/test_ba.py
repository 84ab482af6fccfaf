"""Bundle adjustment hand-off (the reference's ba_demo, bal_example.cpp:44-243).

CPU part: the numpy restatement (oracle/ba_oracle.py) against finite differences and its own
invariants, the BAL reader, the C-ABI's argument checks.  GPU part: the HIP pipeline
(sim3opt_amd/csrc/ba.hip) through the C-ABI against that restatement.

Parity status: "parity unpinned" -- g2o is not in /root/reference (third-party, not vendored) and
the reference holds no ba_demo output, so nothing pins the restatement to g2o beyond the published
formulas it follows (EdgeProjectXYZ2UV::linearizeOplus, SE3Quat::exp, RobustKernelHuber,
OptimizationAlgorithmLevenberg).  The tests below pin the GPU path to the restatement."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ba_oracle as BO  # noqa: E402
from sim3opt_amd import lib as L  # noqa: E402
from tests import kitti_graph as K  # noqa: E402

KF_DIR = os.path.join(K.FIXTURE, "keyframes")


def synthetic_problem(**kw):
    d = BO.synthetic(**kw)
    return BO.Problem(d["cams"], d["points"], d["obs_cam"], d["obs_point"], d["obs_uv"]), d


KF45_DIR = os.path.join(K.FIXTURE, "keyframes45")  # the first 45 keyframes of the reference's KITTI-00 map


def keyframe_problem(kf_dir=KF_DIR):
    """Fixture keyframes as figureKITTIBA hands them to SaveBALFile
    (drawPTAMPoints.cpp:333-371): later keyframes overwrite a point, ids compacted."""
    frames = [L.read_keyframe_bin(os.path.join(kf_dir, n)) for n in sorted(os.listdir(kf_dir))]
    allpts = {}
    for d in frames:
        for pid, p in zip(d["point_ids"], d["points_w"]):
            allpts[int(pid)] = p
    compact = {pid: k for k, pid in enumerate(sorted(allpts))}
    points = np.array([allpts[pid] for pid in sorted(allpts)])
    oc = np.concatenate([np.full(len(d["point_ids"]), c) for c, d in enumerate(frames)])
    op = np.concatenate([[compact[int(p)] for p in d["point_ids"]] for d in frames])
    uv = np.concatenate([d["obs_uv"] for d in frames])
    R = np.array([np.asarray(d["Rw2c"]).reshape(3, 3) for d in frames])
    t = np.array([d["twinc"] for d in frames])
    cams = np.concatenate([BO.R_to_quat(R), t], axis=1)
    return cams, points, oc, op, uv, R, t


# ------------------------------------------------------------------------------------------ CPU
def test_oracle_jacobians_match_central_differences():
    P, _ = synthetic_problem(n_cams=5, n_points=80, seed=3)
    Jp, Jc = P.jacobians()
    h = 1e-6
    rng = np.random.default_rng(0)
    for o in rng.choice(len(P.oc), 12, replace=False):
        c, p = P.oc[o], P.op[o]
        for k in range(6):
            d = np.zeros(6 * len(P.cams) + 3 * len(P.points))
            d[6 * c + k] = h
            ep = P.residuals(*P.apply(P.cams, P.points, d))[o]
            em = P.residuals(*P.apply(P.cams, P.points, -d))[o]
            assert np.abs((ep - em) / (2 * h) - Jc[o][:, k]).max() < 1e-5 * max(1.0, np.abs(Jc[o]).max())
        for k in range(3):
            d = np.zeros(6 * len(P.cams) + 3 * len(P.points))
            d[6 * len(P.cams) + 3 * p + k] = h
            ep = P.residuals(*P.apply(P.cams, P.points, d))[o]
            em = P.residuals(*P.apply(P.cams, P.points, -d))[o]
            assert np.abs((ep - em) / (2 * h) - Jp[o][:, k]).max() < 1e-5 * max(1.0, np.abs(Jp[o]).max())


def test_oracle_se3_exp_is_a_rigid_motion_and_matches_series():
    rng = np.random.default_rng(1)
    u = rng.standard_normal((6, 6)) * 0.3
    u[0, :3] *= 1e-7  # the small-angle branch of se3quat.h
    R, t = BO.se3_exp(u)
    for k in range(6):
        assert np.abs(R[k] @ R[k].T - np.eye(3)).max() < (1e-12 if k else 1e-10)
        # matrix exponential of the 4x4 twist by its series
        X = np.zeros((4, 4))
        X[:3, :3] = BO.skew(u[k, :3][None])[0]
        X[:3, 3] = u[k, 3:]
        E, term = np.eye(4), np.eye(4)
        for n in range(1, 30):
            term = term @ X / n
            E = E + term
        # (small angle: se3quat.h takes V = R = I + Omega + Omega^2 instead of I + Omega / 2 -- kept as written)
        assert np.abs(E[:3, :3] - R[k]).max() < 1e-10 and np.abs(E[:3, 3] - t[k]).max() < (1e-10 if k else 1e-7)


def test_oracle_huber_and_lm_decrease():
    P, d = synthetic_problem(n_cams=6, n_points=150, seed=5)
    e2 = np.array([0.0, 1.0, 6.25, 6.26, 100.0])
    rho, w = P.robust(np.stack([np.sqrt(e2), np.zeros(5)], axis=1))
    assert np.allclose(rho[:3], e2[:3]) and np.allclose(w[:3], 1.0)
    assert np.isclose(rho[4], 2 * 10 * 2.5 - 6.25) and np.isclose(w[4], 0.25)
    c0 = P.chi2()
    tr = P.optimize(6)
    chis = [c0] + [t["chi2"] for t in tr]
    assert all(b <= a for a, b in zip(chis, chis[1:])) and chis[-1] < 0.2 * c0


def test_oracle_schur_solve_equals_full_solve():
    P, _ = synthetic_problem(n_cams=7, n_points=200, seed=9)
    P.fixed[0] = True
    H, b, _ = P.system()
    full, red = P.solve(H, b, 2.5), P.solve(H, b, 2.5, schur=True)
    assert np.abs(full - red).max() < 1e-9 * np.abs(full).max()


def test_bal_reader_matches_writer(tmp_path):
    cams, points, oc, op, uv, R, t = keyframe_problem()
    path = str(tmp_path / "kf.txt")
    L.write_bal(path, R.reshape(-1, 9), t, [718.856, 0, 0], points, oc, op, uv)
    P = BO.read_bal(path)
    assert P.cams.shape == (3, 7) and len(P.points) == len(points) and len(P.oc) == len(oc)
    s = np.sign(np.sum(P.cams[:, :4] * cams[:, :4], axis=1))[:, None]
    assert np.abs(P.cams[:, :4] * s - cams[:, :4]).max() < 1e-12
    assert np.abs(P.cams[:, 4:] - cams[:, 4:]).max() < 1e-12
    b = L.BundleAdjuster()
    b.read_bal(path)
    assert b.dims() == (3, len(points), len(oc))
    got = b.cameras()
    s = np.sign(np.sum(got[:, :4] * P.cams[:, :4], axis=1))[:, None]
    assert np.abs(got[:, :4] * s - P.cams[:, :4]).max() < 1e-14
    assert np.abs(got[:, 4:] - P.cams[:, 4:]).max() == 0 and np.abs(b.points() - P.points).max() == 0
    with pytest.raises(L.Sim3OptError):
        b.read_bal(str(tmp_path / "missing.txt"))
    open(str(tmp_path / "short.txt"), "w").write("2 3 4\n0 0 1 1\n")
    with pytest.raises(L.Sim3OptError):
        b.read_bal(str(tmp_path / "short.txt"))


def test_ba_argument_checks():
    b = L.BundleAdjuster()
    cams = np.tile([0, 0, 0, 1, 0, 0, 0.0], (2, 1))
    pts = np.array([[0, 0, 5.0], [1, 0, 6.0]])
    with pytest.raises(L.Sim3OptError):  # observation of a camera that does not exist (:140-143 asserts)
        b.set_problem(cams, pts, [0, 2], [0, 1], [[1, 1], [2, 2]])
    with pytest.raises(L.Sim3OptError):
        b.set_problem(cams, pts, [0, 1], [0, 5], [[1, 1], [2, 2]])
    with pytest.raises(ValueError):
        b.set_problem(cams, pts, [0, 1], [0], [[1, 1], [2, 2]])
    with pytest.raises(L.Sim3OptError):  # non-finite input
        b.set_problem(cams, [[0, 0, np.nan], [1, 0, 6.0]], [0, 1], [0, 1], [[1, 1], [2, 2]])
    with pytest.raises(L.Sim3OptError):
        b.set_problem(cams, pts, [0, 1], [0, 1], [[1, np.inf], [2, 2]])
    with pytest.raises(L.Sim3OptError):
        b.set_options(pixel_noise=0.0)
    with pytest.raises(AttributeError):
        b.set_options(nonsense=1)
    assert b.optimize(5) == -1  # empty problem: g2o's optimize() returns -1 without edges


# ------------------------------------------------------------------------------------------ GPU
def gpu_problem(P, **opts):
    b = L.BundleAdjuster(**opts)
    b.set_problem(P.cams, P.points, P.oc, P.op, P.uv, P.f, P.cx, P.cy)
    return b


def quat_dist(a, b):
    s = np.sign(np.sum(a * b, axis=1))[:, None]
    return np.abs(a * s - b).max()


@pytest.mark.gpu
@pytest.mark.parametrize("huber", [2.5, 0.0])
def test_gpu_chi2_matches_oracle(huber):
    P, _ = synthetic_problem(n_cams=8, n_points=300, seed=11)
    P.huber = huber
    b = gpu_problem(P, huber_delta=huber)
    want = P.chi2()
    assert abs(b.chi2() - want) <= 1e-12 * want


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_cams,n_points,solver", [(0, 12, 400, 1), (7, 30, 1500, 1), (7, 30, 1500, 0),
                                                         (3, 60, 2500, -1)])
def test_gpu_lm_trace_matches_oracle(seed, n_cams, n_points, solver):
    """solver 1: the reduced camera system by the exact block Cholesky (what ba_demo's LinearSolverEigen
    does); 0: the block-Jacobi PCG fallback; -1: automatic."""
    P, _ = synthetic_problem(n_cams=n_cams, n_points=n_points, seed=seed)
    b = gpu_problem(P, linear_solver=solver)
    n = b.optimize(6)
    tr = P.optimize(6)
    st = b.stats()
    assert n == len(tr) == len(st)
    for s, t in zip(st, tr):
        assert s["trials"] == t["trials"]
        assert abs(s["chi2_after"] - t["chi2"]) <= 1e-7 * t["chi2"]
        assert abs(s["lambda_"] - t["lam"]) <= 1e-5 * t["lam"]
        assert s["pcg_rel_res"] < 1e-10 and (s["pcg_iters"] > 0) == (solver == 0)
    assert quat_dist(b.cameras()[:, :4], P.cams[:, :4]) < 1e-8
    assert np.abs(b.cameras()[:, 4:] - P.cams[:, 4:]).max() < 1e-7
    assert np.abs(b.points() - P.points).max() < 1e-6


@pytest.mark.gpu
def test_gpu_fixed_cameras_keep_their_pose_and_match_oracle():
    """Vertex::setFixed on the first two cameras (the usual gauge anchor; ba_demo itself fixes none)."""
    P, _ = synthetic_problem(n_cams=9, n_points=300, seed=21)
    mask = np.zeros(9, dtype=np.uint8)
    mask[:2] = 1
    P.fixed = mask.astype(bool)
    start = P.cams.copy()
    b = gpu_problem(P)
    b.set_fixed_cameras(mask)
    before = b.cameras()  # (set_problem renormalises the quaternions: last-bit differences from `start`)
    n = b.optimize(5)
    tr = P.optimize(5)
    assert n == len(tr)
    for s, t in zip(b.stats(), tr):
        assert s["trials"] == t["trials"] and abs(s["chi2_after"] - t["chi2"]) <= 1e-7 * t["chi2"]
    got = b.cameras()
    assert np.array_equal(got[:2], before[:2]) and np.abs(before - start).max() < 1e-15 and np.abs(got[2:] - start[2:]).max() > 1e-4
    assert quat_dist(got[:, :4], P.cams[:, :4]) < 1e-8 and np.abs(got[:, 4:] - P.cams[:, 4:]).max() < 1e-7
    with pytest.raises(ValueError):
        b.set_fixed_cameras(mask[:3])


@pytest.mark.gpu
@pytest.mark.parametrize("huber", [2.5, 0.0])
def test_gpu_ragged_problem_matches_oracle(huber):
    """Ragged input: a camera without any observation, a point seen once, a point seen twice by the
    SAME camera (cross terms inside the camera's own block), with and without the robust kernel."""
    d = BO.synthetic(n_cams=8, n_points=250, seed=13)
    cams = np.vstack([d["cams"], d["cams"][-1] + [0, 0, 0, 0, 0.5, 0, 1.0]])  # camera 8: no observations
    pts = np.vstack([d["points"], [[0.3, -0.2, 14.0]]])                        # point seen once
    k = len(pts) - 1
    oc = np.concatenate([d["obs_cam"], [2, 3, 3]])
    op = np.concatenate([d["obs_point"], [k, 5, 5]])                            # point 5: twice by camera 3
    X = BO.quat_to_R(cams[[2, 3, 3], :4]) @ pts[[k, 5, 5], :, None]
    X = X[:, :, 0] + cams[[2, 3, 3], 4:]
    uv_new = np.stack([718.856 * X[:, 0] / X[:, 2] + 607.1928, 718.856 * X[:, 1] / X[:, 2] + 185.2157], axis=1)
    uv = np.vstack([d["obs_uv"], uv_new + [[0.3, -0.4], [1.0, 0.5], [-0.7, 0.2]]])
    P = BO.Problem(cams, pts, oc, op, uv, huber=huber)
    b = gpu_problem(P, huber_delta=huber)
    assert abs(b.chi2() - P.chi2()) <= 1e-12 * P.chi2()
    n = b.optimize(5)
    tr = P.optimize(5)
    assert n == len(tr)
    for s, t in zip(b.stats(), tr):
        assert s["trials"] == t["trials"] and abs(s["chi2_after"] - t["chi2"]) <= 1e-7 * t["chi2"]
    got = b.cameras()
    assert quat_dist(got[:, :4], P.cams[:, :4]) < 1e-8 and np.abs(got[:, 4:] - P.cams[:, 4:]).max() < 1e-7
    # the unobserved camera has a zero gradient: damping alone keeps it where it was
    assert np.abs(got[8] - cams[8] / np.r_[np.ones(4) * np.linalg.norm(cams[8, :4]), np.ones(3)]).max() < 1e-12
    assert np.abs(b.points() - P.points).max() < 1e-6


@pytest.mark.gpu
def test_gpu_single_step_matches_oracle_solve():
    """One LM trial with lambda given: the Schur-complement step equals the oracle's solve of the
    full system (cameras AND points) to solver precision."""
    P, _ = synthetic_problem(n_cams=10, n_points=350, seed=2)
    lam = 3.0
    b = gpu_problem(P, user_lambda_init=lam, max_trials=1)
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    H, rhs, chi = P.system()
    dx = spla.spsolve((H + lam * sp.identity(H.shape[0])).tocsc(), rhs)
    cn, pn = P.apply(P.cams, P.points, dx)
    assert b.optimize(1) == 1
    assert quat_dist(b.cameras()[:, :4], cn[:, :4]) < 1e-11
    assert np.abs(b.cameras()[:, 4:] - cn[:, 4:]).max() < 1e-10
    assert np.abs(b.points() - pn).max() < 1e-9
    s = b.stats()[0]
    assert abs(s["chi2_before"] - chi) <= 1e-12 * chi and abs(s["chi2_after"] - P.chi2(cn, pn)) <= 1e-9 * chi


@pytest.mark.gpu
def test_gpu_kitti_keyframes_ba(tmp_path):
    """ba_demo's flow on real data: the three fixture keyframes -> BAL file -> 5 LM iterations
    (the demo's default, bal_example.cpp:52) -> pose file."""
    cams, points, oc, op, uv, R, t = keyframe_problem()
    path = str(tmp_path / "kf.txt")
    L.write_bal(path, R.reshape(-1, 9), t, [718.856, 0, 0], points, oc, op, uv)
    P = BO.read_bal(path)
    b = L.BundleAdjuster()
    b.read_bal(path)
    c0 = b.chi2()
    assert abs(c0 - P.chi2()) <= 1e-12 * c0
    n = b.optimize(5)
    tr = P.optimize(5)
    st = b.stats()
    assert n == len(tr)
    for s, tt in zip(st, tr):
        assert s["trials"] == tt["trials"] and abs(s["chi2_after"] - tt["chi2"]) <= 1e-6 * tt["chi2"]
    assert st[-1]["chi2_after"] < c0
    assert quat_dist(b.cameras()[:, :4], P.cams[:, :4]) < 1e-7
    assert np.abs(b.cameras()[:, 4:] - P.cams[:, 4:]).max() < 1e-6
    out = str(tmp_path / "poses.txt")
    b.write_poses(out)
    lines = open(out).read().splitlines()
    assert lines[0].startswith("% SE3 optimization result: kf id, tcinw, rc2w(qxyzw)")
    rows = np.array([[float(x) for x in ln.split()] for ln in lines[1:]])
    assert rows.shape == (3, 8) and list(rows[:, 0]) == [0, 1, 2]
    got = b.cameras()
    Rw2c = BO.quat_to_R(got[:, :4])
    tcinw = -np.einsum("nji,nj->ni", Rw2c, got[:, 4:])
    assert np.abs(rows[:, 1:4] - tcinw).max() < 1e-12
    assert quat_dist(rows[:, 4:8], got[:, :4] * np.array([-1, -1, -1, 1.0])) < 1e-15


@pytest.mark.gpu
def test_gpu_real_map_piece_45_keyframes(tmp_path):
    """ba_demo on a connected piece of the reference's real KITTI-00 map: its first 45 keyframes
    (7 802 points, 20 510 observations, tracks of 1..11+ keyframes, depths down to 5 cm), through the
    BAL file, 5 iterations (the demo's default).  The restatement solves the full system by sparse LU
    here, the GPU by Schur complement + block Cholesky."""
    cams, points, oc, op, uv, R, t = keyframe_problem(KF45_DIR)
    assert cams.shape[0] == 45 and len(points) == 7802 and len(oc) == 20510
    path = str(tmp_path / "kf45.bal")
    L.write_bal(path, R.reshape(-1, 9), t, [718.856, 0, 0], points, oc, op, uv)
    P = BO.read_bal(path)
    b = L.BundleAdjuster()
    b.read_bal(path)
    c0 = b.chi2()
    assert abs(c0 - P.chi2()) <= 1e-12 * c0
    n = b.optimize(5)
    tr = P.optimize(5)
    st = b.stats()
    assert n == len(tr) == 5
    for s, tt in zip(st, tr):
        assert s["trials"] == tt["trials"] and abs(s["chi2_after"] - tt["chi2"]) <= 1e-8 * tt["chi2"]
        assert abs(s["lambda_"] - tt["lam"]) <= 1e-6 * tt["lam"]
    assert st[-1]["chi2_after"] < 0.25 * c0
    assert quat_dist(b.cameras()[:, :4], P.cams[:, :4]) < 1e-9
    assert np.abs(b.cameras()[:, 4:] - P.cams[:, 4:]).max() < 1e-8
    assert np.abs(b.points() - P.points).max() < 1e-6 * max(1.0, np.abs(P.points).max())


@pytest.mark.gpu
def test_gpu_ba_recovers_ground_truth_shape():
    """Noise-free observations from a perturbed start: chi2 falls by orders of magnitude."""
    d = BO.synthetic(n_cams=10, n_points=500, seed=4, noise_px=0.0, outliers=0.0)
    P = BO.Problem(d["cams"], d["points"], d["obs_cam"], d["obs_point"], d["obs_uv"])
    b = gpu_problem(P)
    c0 = b.chi2()
    b.optimize(15)
    assert b.chi2() < 1e-6 * c0
