"""The driver's window (bench.py --steps 20 --warmup 5) with and without the adaptive
block-Jacobi-first rule for damping-dominated solves."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sim3opt_amd import lib as L, synth
synth.DRIFT_TARGET = 0.05
g = synth.manhattan()
for adaptive in ("0", "1", "0", "1"):
    os.environ["SIM3OPT_ADAPTIVE_PREC"] = adaptive
    G = L.Graph(fix_small_angle_b=1, pcg_rel_tol=1e-8, time_kernels=1, verbose=int(os.environ.get("VERBOSE", "0")))
    G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"]); G.initialize()
    G.optimize(5); G.set_vertices(g["states"])
    t = time.perf_counter(); n = G.optimize(20); dt = time.perf_counter() - t
    st = G.stats()
    print("adaptive %s: %d it %.1f ms = %.2f LM it/s; first 8: %.1f ms; chi2 %.9f; trials %s pcg %s" % (
        adaptive, n, dt * 1e3, n / dt, sum(s.ms_linearize + s.ms_solve + s.ms_update for s in st[:8]), st[-1].chi2_after,
        [s.trials for s in st], [s.pcg_iters for s in st]), flush=True)
    G.close()
