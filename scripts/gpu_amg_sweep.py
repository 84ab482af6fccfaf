"""Sweep of the multigrid cycle knobs on config 3 (env SIM3OPT_AMG_CYCLE / _ADDITIVE / _OMEGA)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sim3opt_amd import lib as L, synth
V, E, side = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (100000, 1000000, 100)
g = synth.manhattan(V, E, dims=(side, side, 10))
configs = [c.split(":") for c in (sys.argv[4] if len(sys.argv) > 4 else "1:0:0.8,2:0:0.8,12:0:0.8,1:1:0.8,2:1:0.8,122:1:0.8").split(",")]
for cfg in configs:
    cyc, add, om = cfg[:3]
    os.environ["SIM3OPT_AMG_PASSES"] = cfg[3] if len(cfg) > 3 else "3"
    os.environ["SIM3OPT_AMG_COARSEST"] = cfg[4] if len(cfg) > 4 else "256"
    os.environ["SIM3OPT_AMG_CYCLE"] = cyc; os.environ["SIM3OPT_AMG_ADDITIVE"] = add; os.environ["SIM3OPT_AMG_OMEGA"] = om
    G = L.Graph(fix_small_angle_b=1, pcg_rel_tol=1e-8, preconditioner=2)
    G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"]); G.initialize()
    G.optimize(1)
    G.set_vertices(g["states"])
    NIT = int(os.environ.get('NIT', '10'))
    t = time.perf_counter(); G.optimize(NIT); dt = time.perf_counter() - t
    st = G.stats()
    its = [s.pcg_iters for s in st]
    print("levels", G.amg_hierarchy()[0], end=" ")
    print("cycle %-4s additive %s omega %s: 10 LM it %.3fs  chi %.6g  pcg %s  ms/pcg-it %.3f relres %s lam %s" % (
        cyc, add, om, dt, st[-1].chi2_after, its, sum(s.ms_solve for s in st) / max(1, sum(its)), ['%.0e' % s.pcg_rel_res for s in st][-6:], ['%.1e' % s.lambda_ for s in st][-6:]), flush=True)
    G.close()
