"""Config 3 at the noise floor of the delta = 1e-9 Jacobians (LM iterations 10-20 of the driver's
window): per LM iteration, from the same state and the same lambda, the multigrid-preconditioned PCG
against plain block-Jacobi -- which solves are 'easy' (damping-dominated), and what each costs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sim3opt_amd import lib as L, synth
synth.DRIFT_TARGET = 0.05
g = synth.manhattan()
def mk(prec):
    G = L.Graph(fix_small_angle_b=1, pcg_rel_tol=1e-8, preconditioner=prec)
    G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"]); G.initialize()
    return G
A, B = mk(2), mk(0)
A.optimize(2); A.set_vertices(g["states"]); B.optimize(1); B.set_vertices(g["states"])
n0 = int(os.environ.get("N0", "9"))
A.optimize(n0)
lam = A.stats()[-1].lambda_
# mean / max diagonal of H at this state
A.linearize()
rowptr, colidx = A.system_pattern()
for it in range(n0, n0 + int(os.environ.get("NIT", "12"))):
    st = A.get_vertices().copy()
    res = []
    for G in (A, B):
        G.set_vertices(st); G.set_options(user_lambda_init=lam)
        t = time.perf_counter(); G.optimize(1); dt = time.perf_counter() - t
        s = G.stats()[0]
        res.append((dt, s))
    (ta, sa), (tb, sb) = res
    print("it %2d lambda_in %.3e: AMG %6.2f ms trials %d pcg %3d (solve %.2f ms) chi %.6f lam_out %.3e | BJ %6.2f ms trials %d pcg %4d (solve %.2f ms) chi %.6f" % (
        it, lam, ta * 1e3, sa.trials, sa.pcg_iters, sa.ms_solve, sa.chi2_after, sa.lambda_, tb * 1e3, sb.trials, sb.pcg_iters, sb.ms_solve, sb.chi2_after), flush=True)
    A.set_vertices(st); A.set_options(user_lambda_init=lam); A.optimize(1)
    lam = A.stats()[0].lambda_
