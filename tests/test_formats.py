"""Interchange formats and map re-anchoring."""
import os
import struct

import numpy as np
import pytest

from sim3opt_amd import lib as L, sim3np as S3
import kitti_graph as K

KF_DIR = os.path.join(K.FIXTURE, "keyframes")


def read_bin_py(path):
    """Independent Python restatement of LoadComboKeyFrame (drawPTAMPoints.cpp:33-84)."""
    b = open(path, "rb").read()
    o = 0
    kf, sl = struct.unpack_from("<ii", b, o); o += 8 + sl
    o += 16 + 40
    R = np.array(struct.unpack_from("<9d", b, o)).reshape(3, 3); o += 72
    t = np.array(struct.unpack_from("<3d", b, o)); o += 24 + 1
    n, = struct.unpack_from("<i", b, o); o += 4
    ids, pts, uv = [], [], []
    for _ in range(n):
        pid, = struct.unpack_from("<I", b, o); o += 4
        pts.append(struct.unpack_from("<3d", b, o)); o += 24 + 8
        uv.append(struct.unpack_from("<2d", b, o)); o += 16
        ids.append(pid)
    return kf, R, t, np.array(ids, dtype=np.uint32), np.array(pts), np.array(uv)


@pytest.mark.parametrize("name,kf,n", [("KeyFrame000000.bin", 0, 87), ("KeyFrame000011.bin", 11, 436),
                                       ("KeyFrame000012.bin", 12, None)])
def test_keyframe_bin_reader(name, kf, n):
    d = L.read_keyframe_bin(os.path.join(KF_DIR, name))
    kf2, R, t, ids, pts, uv = read_bin_py(os.path.join(KF_DIR, name))
    assert d["kf_id"] == kf == kf2 and (n is None or len(d["point_ids"]) == n)
    assert np.array_equal(d["Rw2c"], R) and np.array_equal(d["twinc"], t)
    assert np.array_equal(d["point_ids"], ids) and np.array_equal(d["points_w"], pts)
    assert np.array_equal(d["obs_uv"], uv)
    assert np.abs(R @ R.T - np.eye(3)).max() < 1e-9
    # image observations are inside the 1241 x 376 KITTI frame (kitti_surf.cpp:52-57)
    assert (uv[:, 0] >= 0).all() and (uv[:, 0] <= 1241).all() and (uv[:, 1] <= 376).all()
    with pytest.raises(L.Sim3OptError) as ei:
        L.read_keyframe_bin(os.path.join(KF_DIR, "missing.bin"))
    assert ei.value.code == L.ERR_IO


def test_g2o_export_roundtrip(tmp_path):
    G = L.Graph()
    G.load_kitti_direct(K.FIXTURE, True)
    path = str(tmp_path / "kitti00.g2o")
    G.write_g2o(path)
    lines = open(path).read().splitlines()
    v = [ln.split() for ln in lines if ln.startswith("VERTEX_SIM3:EXPMAP")]
    e = [ln.split() for ln in lines if ln.startswith("EDGE_SIM3:EXPMAP")]
    assert len(v) == 771 and len(e) == 771 and "FIX 0" in lines
    st = G.get_vertices()
    for row in (v[5], v[400]):  # exp of the written log recovers cam2world = S^-1
        k = int(row[1])
        back = S3.exp(np.array([float(x) for x in row[2:9]]), fix_b=True)
        ref = S3.inv(st[k])
        sg = np.sign(back[:4] @ ref[:4])
        back[:4] *= sg
        assert np.abs(back - ref).max() < 1e-9
    a, b, m = G.get_edge(0)
    assert (int(e[0][1]), int(e[0][2])) == (a, b) and len(e[0]) == 3 + 7 + 28


@pytest.mark.gpu
def test_map_reanchoring_matches_sequential_restatement():
    """drawPTAMPoints.cpp:416-429 on the GPU vs the literal sequential loop, on the three vendored
    keyframes (real point ids / coordinates) and on a synthetic correction."""
    kfs = [L.read_keyframe_bin(os.path.join(KF_DIR, f)) for f in
           ("KeyFrame000000.bin", "KeyFrame000011.bin", "KeyFrame000012.bin")]
    all_ids = np.unique(np.concatenate([k["point_ids"] for k in kfs]))
    cid = {int(p): i for i, p in enumerate(all_ids)}
    points = np.zeros((len(all_ids), 3))
    obs_f, obs_p = [], []
    for f, k in enumerate(kfs):  # later keyframes overwrite the stored point (drawPTAMPoints.cpp:334-345)
        for pid, pw in zip(k["point_ids"], k["points_w"]):
            points[cid[int(pid)]] = pw
            obs_f.append(f)
            obs_p.append(cid[int(pid)])
    old_Rt = np.array([np.concatenate([k["Rw2c"].ravel(), k["twinc"]]) for k in kfs])
    rng = np.random.default_rng(3)
    new = np.array([S3.mul(S3.exp(np.concatenate([rng.standard_normal(3) * 0.02,
                                                  rng.standard_normal(3) * 0.1,
                                                  [rng.standard_normal() * 0.1]]), fix_b=True),
                           S3.make(S3.R_to_quat(k["Rw2c"]), k["twinc"], 1.0)) for k in kfs])
    got = L.reanchor_points(old_Rt, new, points, obs_f, obs_p)
    want = points.copy()
    for f, p in zip(obs_f, obs_p):  # the reference's loop, verbatim semantics
        rel = old_Rt[f, :9].reshape(3, 3) @ points[p] + old_Rt[f, 9:]
        Si = S3.inv(new[f])
        want[p] = Si[7] * (S3.quat_to_R(Si[:4]) @ rel) + Si[4:7]
    assert np.abs(got - want).max() < 1e-9
    # identity correction leaves the map where it was
    same = np.array([S3.make(S3.R_to_quat(k["Rw2c"]), k["twinc"], 1.0) for k in kfs])
    assert np.abs(L.reanchor_points(old_Rt, same, points, obs_f, obs_p) - points).max() < 1e-9


def test_bal_writer_from_keyframes(tmp_path):
    """SaveBALFile (drawPTAMPoints.cpp:218-283) on the three fixture keyframes, with the id
    compaction of figureKITTIBA (:347-371): parse the file back and compare."""
    frames = [L.read_keyframe_bin(os.path.join(KF_DIR, n)) for n in sorted(os.listdir(KF_DIR))]
    allpts = {}
    for d in frames:  # later keyframes overwrite a point's coordinates (:333-341)
        for pid, p in zip(d["point_ids"], d["points_w"]):
            allpts[int(pid)] = p
    compact = {pid: k for k, pid in enumerate(sorted(allpts))}
    points = np.array([allpts[pid] for pid in sorted(allpts)])
    oc = np.concatenate([np.full(len(d["point_ids"]), c) for c, d in enumerate(frames)])
    op = np.concatenate([[compact[int(p)] for p in d["point_ids"]] for d in frames])
    uv = np.concatenate([d["obs_uv"] for d in frames])
    R = np.array([d["Rw2c"] for d in frames])
    t = np.array([d["twinc"] for d in frames])
    path = str(tmp_path / "problem.txt")
    L.write_bal(path, R, t, [718.856, 0, 0], points, oc, op, uv)
    tok = open(path).read().split()
    nc, npnt, no = int(tok[0]), int(tok[1]), int(tok[2])
    assert (nc, npnt, no) == (3, len(points), len(oc))
    o = 3
    obs = np.array(tok[o:o + 4 * no], dtype=np.float64).reshape(no, 4); o += 4 * no
    assert np.array_equal(obs[:, 0], oc) and np.array_equal(obs[:, 1], op)
    assert np.abs(obs[:, 2:] - uv).max() < 5e-4 * np.abs(uv).max()  # %g: 6 significant digits
    cams = np.array(tok[o:o + 9 * nc], dtype=np.float64).reshape(nc, 9); o += 9 * nc
    for c in range(nc):  # Rodrigues of the angle-axis gives R_w2c back
        w = cams[c, :3]
        th = np.linalg.norm(w)
        Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]) / max(th, 1e-300)
        Rb = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
        assert np.abs(Rb - R[c].reshape(3, 3)).max() < 1e-12
        # (%.16g as in the reference: one digit short of a bit-exact round trip)
        assert np.abs(cams[c, 3:6] - t[c]).max() <= 1e-15 * np.abs(t[c]).max() and list(cams[c, 6:]) == [718.856, 0, 0]
    pts = np.array(tok[o:], dtype=np.float64).reshape(npnt, 3)
    assert np.abs(pts - points).max() <= 1e-15 * np.abs(points).max()
    with pytest.raises(L.Sim3OptError):  # point ids with a gap: the reference exits (:243-247)
        L.write_bal(path, R, t, [718.856, 0, 0], points, oc, op + (op > 5), uv)
    with pytest.raises(L.Sim3OptError):
        L.write_bal(str(tmp_path / "no" / "dir.txt"), R, t, [1, 0, 0], points, oc, op, uv)
