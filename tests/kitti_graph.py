"""Independent numpy restatement of the reference's KITTI-00 pose-graph construction.

Test helper: builds the arrays (states, fixed, v0, v1, meas) that
testDirectSim3Optimization hands to g2o (kitti_surf.cpp:562-670), from the
vendored input fixture tests/golden/kitti00.  The product's own C++ loader
(sim3opt_amd/csrc/kitti_io.cpp) is checked against this.
"""
import os

import numpy as np

from sim3opt_amd import sim3np as S3

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kitti00")


def load_cc(path=None):
    """kitti_surf.cpp:232-254 -- keyframe id = line number, image id = value."""
    path = path or os.path.join(FIXTURE, "cc.txt")
    return [int(x) for x in open(path).read().split()]


def load_kf_poses(cc, path=None):
    """kitti_surf.cpp:255-292 -- returns (771, 8) S_iw = (R_w2c, t_w2c, 1)."""
    path = path or os.path.join(FIXTURE, "framePoses_kf.txt")
    want = set(cc)
    rows = {}
    with open(path) as f:
        lines = f.read().splitlines()[2:]
    for ln in lines:
        if not ln.strip():
            continue
        p = [x.strip() for x in ln.split(",")]
        fid = int(p[0])
        if fid in want:
            rows[fid] = [float(x) for x in p[2:8]]
    out = np.empty((len(cc), 8))
    for k, fid in enumerate(cc):
        r, p, y, tx, ty, tz = rows[fid]
        Rc2w = S3.euler_rpy_to_R(r, p, y)
        # Sophus SE3(R,t).inverse(): unit quaternion conj, t' = R^-1 * (-t)   (:283-285)
        qc2w = S3.R_to_quat(Rc2w)
        qw2c = S3.quat_conj(qc2w)
        tw2c = S3.quat_rot(qw2c, -np.array([tx, ty, tz]))
        # g2o::Sim3(Rcw, tcw, 1.0) with Rcw = Tw2c.rotationMatrix()      (:606-608)
        q = S3.R_to_quat(S3.quat_to_R(qw2c))
        out[k] = np.concatenate([q, tw2c, [1.0]])
    return out


def load_loop_constraints(path=None):
    """kitti_surf.cpp:145-205 -- list of (frame1, frame2, Sim3 state) from line 1 / line 4."""
    path = path or os.path.join(FIXTURE, "loopConstraints.txt")
    with open(path) as f:
        lines = f.read().splitlines()[5:]
    lines = [ln for ln in lines if ln.strip()]
    out = []
    for k in range(0, len(lines) - 3, 4):
        a = lines[k].split()
        d = lines[k + 3].split()
        f1, f2 = int(a[0]), int(a[1])
        s = float(d[1])
        r, p, y, tx, ty, tz = [float(x) for x in d[2:8]]
        assert r != 0 and p != 0 and y != 0  # kitti_surf.cpp:190
        q = S3.R_to_quat(S3.euler_rpy_to_R(r, p, y))
        out.append((f1, f2, np.concatenate([q, [tx, ty, tz], [s]])))
    return out


def build_direct_graph(use_one_constraint=True):
    """Arrays of the graph of testDirectSim3Optimization (kitti_surf.cpp:575-670).

    Returns dict(states, fixed, v0, v1, meas, image_ids).  Edge order = loop edges
    first, then odometry edges, exactly as the reference adds them.
    """
    cc = load_cc()
    states = load_kf_poses(cc)
    loops = load_loop_constraints()
    if use_one_constraint:
        loops = loops[:1]  # kitti_surf.cpp:568-573
    fid2kf = {fid: k for k, fid in enumerate(cc)}
    v0, v1, meas = [], [], []
    for f1, f2, Cm in loops:  # :624-640  setVertex(0, id1), setVertex(1, id2)
        v0.append(fid2kf[f1])
        v1.append(fid2kf[f2])
        meas.append(Cm)
    for i in range(1, len(cc)):  # :649-670  Sji = Sjw * Swi, setVertex(0, i), setVertex(1, j=i-1)
        Swi = S3.inv(states[i])
        Sji = S3.mul(states[i - 1], Swi)
        v0.append(i)
        v1.append(i - 1)
        meas.append(Sji)
    fixed = np.zeros(len(cc), dtype=np.uint8)
    fixed[0] = 1  # :613-616
    return dict(states=states, fixed=fixed, v0=np.array(v0, dtype=np.int32),
                v1=np.array(v1, dtype=np.int32), meas=np.array(meas), image_ids=cc)
