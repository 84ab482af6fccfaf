"""Config 3 in the reference's own arithmetic (B as written, delta = 1e-9): what the PCG does with
block-Jacobi (the automatic choice so far) and with the multigrid hierarchy forced; verbose output
shows where a set-up pivot fails.  NIT LM iterations (default 12)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sim3opt_amd import lib as L, synth
synth.DRIFT_TARGET = 0.05
g = synth.manhattan(int(os.environ.get("V", 100000)), int(os.environ.get("E", 1000000)))
NIT = int(os.environ.get("NIT", "12"))
for prec in [int(x) for x in (sys.argv[1:] or ["2", "0"])]:
    G = L.Graph(fix_small_angle_b=0, pcg_rel_tol=1e-8, preconditioner=prec, verbose=int(os.environ.get("VERBOSE", "1")))
    G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"]); G.initialize()
    t = time.perf_counter(); n = G.optimize(NIT); dt = time.perf_counter() - t
    st = G.stats()
    print("prec %d (in use %d): %d it in %.2fs = %.2f LM it/s chi %.6g -> %.6g" % (prec, G.preconditioner_in_use(), n, dt, n / dt, st[0].chi2_before, st[-1].chi2_after))
    print("  pcg", [s.pcg_iters for s in st], "rel", ["%.1e" % s.pcg_rel_res for s in st], "trials", [s.trials for s in st], "lambda", ["%.2e" % s.lambda_ for s in st], flush=True)
    G.close()
