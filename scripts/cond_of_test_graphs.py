"""Condition numbers of the LM systems behind the self-comparison bounds of the -m gpu tests:
cond(H + lambda I) at the state and damping the oracle reaches after the test's LM iterations, on the CPU
(oracle LM for the states, dense eigenvalues).  Two runs that solve the same systems to a PCG tolerance tol by
different summation orders / preconditioners differ by up to ~ tol * cond in the step, which is what the bounds
next to the numbers printed here allow.   Usage: python scripts/cond_of_test_graphs.py  -> profiles/r4_condition_numbers.json"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O
from sim3opt_amd import synth

def cond_after(g, iters, fd, name, tol):
    o = O.default_options(fix_small_angle_b=1, fd_delta=fd, threads=8)
    OG = O.Graph(g["states"], g["fixed"], g["v0"], g["v1"], g["meas"])
    t = time.time()
    it, tr = OG.optimize(iters, o)
    lam = tr[-1].lambda_ if hasattr(tr[-1], "lambda_") else tr[-1].lam
    H, b = OG.build_dense(o)
    w = np.linalg.eigvalsh(H)
    out = dict(graph=name, n=int(H.shape[0]), lm_iterations=int(it), lambda_after=float(lam), eig_min_H=float(w[0]),
               eig_max_H=float(w[-1]), cond_H_plus_lambda=float((w[-1] + lam) / (w[0] + lam)), pcg_rel_tol=tol,
               tol_times_cond=float(tol * (w[-1] + lam) / (w[0] + lam)), seconds=time.time() - t)
    print(json.dumps(out), flush=True)
    return out

synth.DRIFT_TARGET = 0.05
res = []
res.append(cond_after(synth.manhattan(300, 2500, dims=(7, 7, 4), per_cell=4), 4, 1e-6,
                      "manhattan 300 / 2500 (test_distributed_gpu: block-Jacobi cases, RCCL self-test)", 1e-12))
res.append(cond_after(synth.manhattan(1500, 15000, dims=(12, 12, 10)), 4, 1e-6,
                      "manhattan 1500 / 15000 (three-level multigrid: test_distributed_gpu prec = 2, "
                      "test_three_level_multigrid_lm_matches_exact_oracle)", 1e-12))
json.dump(res, open(os.path.join(ROOT, "profiles", "r4_condition_numbers.json"), "w"), indent=1)
