"""GPU A/B of the PCG preconditioners on Manhattan graphs: block-Jacobi (0) vs aggregation
multigrid (2).  One linear solve at two dampings (solutions compared), then LM runs.
Usage: python scripts/gpu_amg_run.py [small|full]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sim3opt_amd import lib as L, synth

mode = sys.argv[1] if len(sys.argv) > 1 else "small"
cases = [(6000, 60000, (24, 24, 10))] if mode == "small" else [(6000, 60000, (24, 24, 10)), (100000, 1000000, (100, 100, 10))]
for V, E, dims in cases:
    g = synth.manhattan(V, E, dims=dims)
    print("== Manhattan", V, E, flush=True)
    sol = {}
    for pre in (0, 2):
        G = L.Graph(fix_small_angle_b=1, pcg_rel_tol=1e-10, pcg_max_iters=20000, preconditioner=pre, verbose=0)
        G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"]); G.initialize(); G.linearize()
        rowptr, colidx, blocks, b = G.get_system() if V <= 6000 else (None, None, None, None)
        for lam in (10.0, 1e-2):
            G.solve(lam)  # warm-up (graph capture, hierarchy numbers)
            t = time.perf_counter(); x, it, rr = G.solve(lam); dt = time.perf_counter() - t
            sol[(pre, lam)] = x
            msg = "pre %d lam %-6g iters %5d relres %.1e ms %8.2f" % (pre, lam, it, rr, 1e3 * dt)
            if pre == 2:
                ref = sol[(0, lam)]
                msg += "  |x - x_bj| / |x| %.1e" % (np.abs(x - ref).max() / np.abs(ref).max())
            print(msg, flush=True)
        G.close()
    for pre in (0, 2):
        G = L.Graph(fix_small_angle_b=1, pcg_rel_tol=1e-8, preconditioner=pre, verbose=0)
        G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"]); G.initialize()
        G.optimize(1)
        t = time.perf_counter(); n = G.optimize(10); dt = time.perf_counter() - t
        st = G.stats()
        print("LM pre %d: 10 it %.3fs  chi %s  pcg %s  relres %s" % (
            pre, dt, ["%.5g" % s.chi2_after for s in st], [s.pcg_iters for s in st],
            ["%.0e" % s.pcg_rel_res for s in st]), flush=True)
        G.close()
