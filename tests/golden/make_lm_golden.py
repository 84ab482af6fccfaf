"""Generates tests/golden/kitti_lm_golden.json: an INDEPENDENT restatement of the reference's LM run on
the KITTI-00 graphs, used to pin the CPU oracle (oracle/sim3_oracle.c) and, through it, the HIP path.

What is independent of the oracle here (SURVEY.md 8c plan items iii / iv):
  * the Sim(3) arithmetic: numpy (sim3opt_amd/sim3np.py follows sim3_rv.h:125-190, :242-320),
  * the numeric Jacobians: vectorised central differences written here (g2o BaseBinaryEdge, delta 1e-9),
  * the assembly: scipy.sparse COO -> CSC of the scalar normal equations,
  * the linear solve: scipy.sparse.linalg.spsolve (SuperLU, COLAMD) instead of the oracle's own
    block-minimum-degree LDL^T,
  * the LM policy: restated here from SURVEY.md App. C (OptimizationAlgorithmLevenberg).
The reference itself cannot be built (g2o, vio_g2o, Eigen are absent and unfetchable) and stores no
outputs, so parity with g2o stays unpinned; this file removes the "same author, same code" objection
from the oracle's LM trace.

Two kinds of records per graph (one loop = kitti_surf.cpp:1317 bUseOneContraint; all 118 loops):
  free    the independent LM run freely for N iterations: chi2 / lambda / trials per iteration and
          the final translations t(S_wi).  The reference's configuration is chaotic (the as-written B
          coefficient of sim3_rv.h:166/:290 makes Jacobian entries of 1e6..1e7 after the first
          update; last-bit differences of libm grow tenfold per iteration), so two correct
          implementations agree on this trace only for the first iterations -- the record says how far
          the oracle follows it.
  lockstep  iteration k restarted from the ORACLE's state k and lambda k (every LM iteration ends
          with nu = 2, so lambda and the estimates are the whole state): one iteration of the
          independent LM.  This pins every step of the oracle without the chaos of a free run.

    python tests/golden/make_lm_golden.py        (about two minutes)
"""
import json
import os
import sys

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from sim3opt_amd import sim3np as S3  # noqa: E402
import kitti_graph as K  # noqa: E402

N_FREE = 20
N_LOCK = 20


def edge_errors(g, states):
    return S3.edge_error(g["meas"], states[g["v0"]], states[g["v1"]])


def numeric_jacobians(g, states, delta):
    """(A, B), each (ne, 7, 7) [k, r, c]: central differences through the vertex update exp(d) * S."""
    ne = g["v0"].shape[0]
    A = np.empty((ne, 7, 7))
    B = np.empty((ne, 7, 7))
    S0, S1 = states[g["v0"]], states[g["v1"]]
    for d in range(7):
        xi = np.zeros(7)
        xi[d] = delta
        Pp, Pm = S3.exp(xi), S3.exp(-xi)
        ep = S3.edge_error(g["meas"], S3.mul(Pp, S0), S1)
        em = S3.edge_error(g["meas"], S3.mul(Pm, S0), S1)
        A[:, :, d] = (ep - em) / (2 * delta)
        ep = S3.edge_error(g["meas"], S0, S3.mul(Pp, S1))
        em = S3.edge_error(g["meas"], S0, S3.mul(Pm, S1))
        B[:, :, d] = (ep - em) / (2 * delta)
    return A, B


def assemble(g, states, delta):
    """Scalar normal equations of the free vertices (ascending id = g2o hessianIndex), CSC."""
    fixed = g["fixed"].astype(bool)
    hidx = -np.ones(len(fixed), dtype=np.int64)
    hidx[~fixed] = np.arange((~fixed).sum())
    n = 7 * int((~fixed).sum())
    e = edge_errors(g, states)
    A, B = numeric_jacobians(g, states, delta)
    h0, h1 = hidx[g["v0"]], hidx[g["v1"]]
    rows, cols, vals = [], [], []
    b = np.zeros(n)
    r7 = np.arange(7)

    def add(hr, hc, M, sel):
        rr = (7 * hr[sel])[:, None, None] + r7[None, :, None]
        cc = (7 * hc[sel])[:, None, None] + r7[None, None, :]
        rows.append(np.broadcast_to(rr, M[sel].shape).ravel())
        cols.append(np.broadcast_to(cc, M[sel].shape).ravel())
        vals.append(M[sel].ravel())

    f0, f1 = h0 >= 0, h1 >= 0
    add(h0, h0, np.einsum("kri,krj->kij", A, A), f0)
    add(h1, h1, np.einsum("kri,krj->kij", B, B), f1)
    both = f0 & f1
    AB = np.einsum("kri,krj->kij", A, B)
    add(h0, h1, AB, both)
    add(h1, h0, AB.transpose(0, 2, 1), both)
    np.add.at(b, (7 * h0[f0])[:, None] + r7, -np.einsum("kri,kr->ki", A, e)[f0])
    np.add.at(b, (7 * h1[f1])[:, None] + r7, -np.einsum("kri,kr->ki", B, e)[f1])
    H = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                      shape=(n, n)).tocsc()
    return H, b, float((e * e).sum()), hidx


def apply_step(g, states, hidx, x):
    out = states.copy()
    free = np.where(hidx >= 0)[0]
    out[free] = S3.mul(S3.exp(x.reshape(-1, 7)[hidx[free]]), states[free])
    return out


def lm(g, states, iters, lam=None, delta=1e-9, tau=1e-5, max_trials=10):
    """g2o OptimizationAlgorithmLevenberg.  Returns (states, trace)."""
    trace = []
    states = states.copy()
    for it in range(iters):
        H, b, chi_cur, hidx = assemble(g, states, delta)
        if lam is None:
            lam = tau * float(np.abs(H.diagonal()).max())
        ni = 2.0
        rho, q = 0.0, 0
        I = sp.identity(H.shape[0], format="csc")
        while True:
            x = spla.spsolve(H + lam * I, b)
            new = apply_step(g, states, hidx, x)
            e = edge_errors(g, new)
            chi_new = float((e * e).sum())
            scale = float(x @ (lam * x + b)) + 1e-3
            rho = (chi_cur - chi_new) / scale
            if rho > 0 and np.isfinite(chi_new):
                alpha = min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0)
                lam *= max(1.0 / 3.0, alpha)
                ni = 2.0
                states, chi_cur = new, chi_new
            else:
                lam *= ni
                ni *= 2.0
            q += 1
            if not (rho < 0 and q < max_trials):
                break
        trace.append(dict(chi2=chi_cur, lam=lam, trials=q, rho=rho))
        if q == max_trials or rho == 0 or not np.isfinite(lam):
            break
    return states, trace


def main():
    from oracle import oracle as O  # only for the lock-step records: the oracle's own states / lambdas

    out = dict(n_free=N_FREE, n_lock=N_LOCK, delta=1e-9,
               note="independent numpy/scipy LM (make_lm_golden.py); t_wi = translations of S_wi")
    for name, one in (("one_loop", True), ("all_loops", False)):
        g = K.build_direct_graph(one)
        st, tr = lm(g, g["states"], N_FREE)
        rec = dict(free=dict(chi2=[t["chi2"] for t in tr], lam=[t["lam"] for t in tr],
                             trials=[t["trials"] for t in tr],
                             t_wi=S3.inv(st)[:, 4:7].tolist(), scale=st[:, 7].tolist()))
        # lock-step: one independent iteration from each of the oracle's states
        OG = O.Graph(g["states"], g["fixed"], g["v0"], g["v1"], g["meas"])
        lock = []
        lam = None
        for k in range(N_LOCK):
            s_in = OG.states.copy()
            _, t1 = lm(g, s_in, 1, lam=lam)
            o = O.default_options(user_lambda_init=lam if lam is not None else 0.0)
            it, otr = OG.optimize(1, o)
            assert it == 1
            lock.append(dict(lam_in=lam, chi2=t1[0]["chi2"], lam=t1[0]["lam"], trials=t1[0]["trials"],
                             oracle_chi2=otr[0].chi2_after, oracle_lam=otr[0].lambda_,
                             oracle_trials=otr[0].trials))
            lam = otr[0].lambda_
            print(name, "lock", k, lock[-1], flush=True)
        rec["lockstep"] = lock
        # how far the oracle's own free run follows the independent one (recorded, asserted in tests)
        OG = O.Graph(g["states"], g["fixed"], g["v0"], g["v1"], g["meas"])
        it, otr = OG.optimize(N_FREE)
        rec["oracle_free"] = dict(chi2=[t.chi2_after for t in otr], lam=[t.lambda_ for t in otr],
                                  trials=[t.trials for t in otr])
        for k in range(min(len(tr), len(otr))):
            print(name, "free", k, "indep %.10g %.4g %d | oracle %.10g %.4g %d" % (
                tr[k]["chi2"], tr[k]["lam"], tr[k]["trials"], otr[k].chi2_after, otr[k].lambda_,
                otr[k].trials), flush=True)
        out[name] = rec
    with open(os.path.join(HERE, "kitti_lm_golden.json"), "w") as f:
        json.dump(out, f)
    print("wrote kitti_lm_golden.json")


if __name__ == "__main__":
    main()
