"""ba_demo at the size of the reference's KITTI-00 map (771 keyframes, ~123 k points, ~338 k
observations: SURVEY.md section 6) on a synthetic problem of that shape: GPU pipeline (C-ABI) next to
the numpy/scipy restatement, same LM trace expected.  Usage: python scripts/gpu_ba_scale.py [iters] [--no-cpu]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ba_oracle as BO  # noqa: E402  (checker only)
from sim3opt_amd import lib as L  # noqa: E402


def kitti_like(n_cams=771, n_points=123000, seed=0, noise_px=0.5, outliers=0.02):
    rng = np.random.default_rng(seed)
    f, cx, cy = L.KITTI_FOCAL, L.KITTI_CX, L.KITTI_CY
    step = 1.2
    cams = np.zeros((n_cams, 7))
    Rs, ts, pcs = [], [], []
    for c in range(n_cams):
        yaw = 0.004 * c
        Rc2w = np.array([[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0], [-np.sin(yaw), 0, np.cos(yaw)]])
        pc = np.array([40 * np.sin(0.01 * c), 0.0, step * c])
        Rw2c = Rc2w.T
        Rs.append(Rw2c); ts.append(-Rw2c @ pc); pcs.append(pc)
        cams[c, :4] = BO.R_to_quat(Rw2c[None])[0]
        cams[c, 4:] = ts[-1]
    # every point: a first camera and a track of 2..5 consecutive keyframes (mean 2.75 as KITTI-00)
    c0 = rng.integers(0, n_cams - 5, n_points)
    track = rng.choice([2, 2, 2, 3, 3, 4, 5], n_points)
    Xc = np.stack([rng.uniform(-0.5, 0.5, n_points), rng.uniform(-0.12, 0.2, n_points), np.ones(n_points)], axis=1)
    depth = rng.uniform(8, 45, n_points)
    Xc *= depth[:, None]
    pts = np.empty((n_points, 3))
    for c in range(n_cams):
        m = c0 == c
        pts[m] = (Xc[m] - ts[c]) @ Rs[c]  # R^T (X - t)
    oc, op, uv = [], [], []
    for k in range(5):
        m = track > k
        idx = np.where(m)[0]
        cc = c0[idx] + k
        R = np.array(Rs)[cc]
        t = np.array(ts)[cc]
        X = np.einsum("nij,nj->ni", R, pts[idx]) + t
        ok = X[:, 2] > 2
        u = f * X[:, 0] / X[:, 2] + cx
        v = f * X[:, 1] / X[:, 2] + cy
        oc.append(cc[ok]); op.append(idx[ok]); uv.append(np.stack([u, v], axis=1)[ok])
    oc, op, uv = np.concatenate(oc), np.concatenate(op), np.concatenate(uv)
    uv = uv + rng.standard_normal(uv.shape) * noise_px
    bad = rng.random(len(uv)) < outliers
    uv[bad] += rng.standard_normal((int(bad.sum()), 2)) * 40.0
    cams0 = cams.copy()
    cams0[:, 4:] += rng.standard_normal((n_cams, 3)) * 0.03
    pts0 = pts + rng.standard_normal(pts.shape) * 0.15
    return cams0, pts0, oc, op, uv


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 5
    cams, pts, oc, op, uv = kitti_like()
    print(f"problem: {len(cams)} cameras, {len(pts)} points, {len(oc)} observations", flush=True)
    b = L.BundleAdjuster()
    b.set_problem(cams, pts, oc, op, uv)
    c0 = b.chi2()  # includes the one-time structure build + upload
    t0 = time.perf_counter()
    n = b.optimize(iters)
    gpu_s = time.perf_counter() - t0
    st = b.stats()
    print(f"GPU: chi2 {c0:.6f} -> {st[-1]['chi2_after']:.6f}, {n} iterations in {gpu_s:.3f} s; "
          f"trials {[s['trials'] for s in st]}, pcg {[s['pcg_iters'] for s in st]}", flush=True)
    out = {"n_cams": len(cams), "n_points": len(pts), "n_obs": len(oc), "iters": n, "gpu_s": gpu_s,
           "gpu_chi2": [c0] + [s["chi2_after"] for s in st], "gpu_trials": [s["trials"] for s in st],
           "gpu_pcg_iters": [s["pcg_iters"] for s in st]}
    if "--no-cpu" not in sys.argv:
        P = BO.Problem(cams, pts, oc, op, uv)
        t0 = time.perf_counter()
        tr = P.optimize(iters, schur=True)
        cpu_s = time.perf_counter() - t0
        print(f"CPU restatement: -> {tr[-1]['chi2']:.6f} in {cpu_s:.2f} s; trials {[t['trials'] for t in tr]}", flush=True)
        rel = max(abs(s["chi2_after"] - t["chi2"]) / t["chi2"] for s, t in zip(st, tr))
        dq = np.abs(np.abs(np.sum(b.cameras()[:, :4] * P.cams[:, :4], axis=1)) - 1).max()
        dt = np.abs(b.cameras()[:, 4:] - P.cams[:, 4:]).max()
        print(f"max rel chi2 difference {rel:.2e}; cameras: |1-|q.q'|| {dq:.2e}, translation {dt:.2e} m", flush=True)
        out.update({"cpu_s": cpu_s, "cpu_chi2": [t["chi2"] for t in tr], "cpu_trials": [t["trials"] for t in tr],
                    "max_rel_chi2_diff": rel, "max_translation_diff_m": dt})
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "ba_scale.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
