"""Generates tests/golden/oracle_golden.json.

The reference cannot be built or run in this environment (g2o, vio_g2o, Eigen, Sophus are not
in /root/reference and cannot be fetched) and it stores no outputs, so these vectors come from
the CPU oracle (oracle/sim3_oracle.c) and are cross-checked HERE, at generation time, against
the independent numpy restatement sim3opt_amd/sim3np.py.  Parity therefore stays "unpinned"
against g2o itself; the vectors pin the oracle (and through it the HIP path) against silent
drift.  Inputs: the vendored reference data files in tests/golden/kitti00 and seeded synthetic
graphs.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle as O  # noqa: E402
from sim3opt_amd import sim3np as S3, synth  # noqa: E402
import kitti_graph as K  # noqa: E402

out = {}

# ---- exp / log on every branch, just inside / outside the 1e-5 thresholds ----
rng = np.random.default_rng(7)
cases = []
for sig in (0.0, 3e-6, 9.9e-6, 1.01e-5, 1e-3, -0.4, 1.7):
    for th in (0.0, 5e-6, 9.9e-6, 1.01e-5, 4.0e-3, 5.0e-3, 0.3, 2.5):
        ax = rng.standard_normal(3)
        ax /= np.linalg.norm(ax)
        cases.append(np.concatenate([ax * th, rng.standard_normal(3) * 2.0, [sig]]))
cases = np.array(cases)
exp_out = np.array([O.sim3_exp(x) for x in cases])
log_out = np.array([O.sim3_log(s) for s in exp_out])

def _rel(a, b):  # the as-written B coefficient makes some outputs ~1e5, so compare relatively
    return (np.abs(a - b) / (1.0 + np.abs(b))).max()


assert _rel(exp_out, S3.exp(cases)) < 1e-12, "oracle exp disagrees with numpy"
assert _rel(log_out, S3.log(exp_out)) < 1e-8, "oracle log disagrees with numpy"
out["explog"] = dict(xi=cases.tolist(), exp=exp_out.tolist(), log_of_exp=log_out.tolist())

# ---- KITTI-00 (reference data): chi2_0, residuals, Jacobians of a few edges, LM head ----
kit = {}
for name, one in (("one_loop", True), ("all_loops", False)):
    g = K.build_direct_graph(one)
    G = O.Graph(g["states"], g["fixed"], g["v0"], g["v1"], g["meas"])
    e = G.errors()
    e_np = S3.edge_error(g["meas"], g["states"][g["v0"]], g["states"][g["v1"]])
    assert np.abs(e - e_np).max() < 1e-10
    A, B = G.jacobians(O.default_options(fd_delta=1e-6))
    nl = 1 if one else 118
    sel = sorted(set([0, nl - 1, nl, nl + 1, nl + 100, nl + 400, nl + 769] +
                     ([5, 17, 60, 117] if not one else [])))
    it, tr = G.optimize(4)
    kit[name] = dict(
        n_vertices=int(G.nv), n_edges=int(G.ne), chi2_0=float(np.sum(e * e)),
        edge_sel=sel, e_sel=e[sel].tolist(), A_sel_fd1e6=A[sel].tolist(),
        B_sel_fd1e6=B[sel].tolist(),
        lm_chi2_head=[t.chi2_after for t in tr], lm_trials_head=[t.trials for t in tr],
    )
out["kitti"] = kit

# ---- small seeded synthetic graphs, well-posed mode (fix_small_angle_b = 1) ----
# two finite-difference steps: g2o's 1e-9 (Jacobian noise ~1e-7, LM stalls at that floor) and
# 1e-6 (noise ~1e-10: both implementations reach the same optimum to ~1e-7)
syn = {}
synth.DRIFT_TARGET = 0.05
for name, g in (("manhattan_120", synth.manhattan(120, 1000, dims=(6, 6, 3), per_cell=4)),
                ("chain_150", synth.chain_loop(150, 300))):
    rec = {}
    for tag, fd in (("fd1e9", 1e-9), ("fd1e6", 1e-6)):
        G = O.Graph(g["states"], g["fixed"], g["v0"], g["v1"], g["meas"])
        o = O.default_options(fix_small_angle_b=1, fd_delta=fd)
        chi0 = G.chi2(o)
        it, tr = G.optimize(15, o)
        rec[tag] = dict(chi2_0=chi0, iters=it, chi2_final=tr[-1].chi2_after,
                        positions=synth.positions(G.states).tolist(),
                        scales=G.states[:, 7].tolist())
    d = np.array(rec["fd1e9"]["positions"]) - np.array(rec["fd1e6"]["positions"])
    rec["rmse_between_fd_steps"] = float(np.sqrt((d ** 2).sum(1).mean()))
    syn[name] = rec
out["synthetic_fixb"] = syn

with open(os.path.join(HERE, "oracle_golden.json"), "w") as f:
    json.dump(out, f)
print("wrote oracle_golden.json", {k: (list(v.keys()) if isinstance(v, dict) else len(v))
                                  for k, v in out.items()})
