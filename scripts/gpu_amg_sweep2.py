"""Round-2 sweep of the multigrid knobs on config 3 (one process, graph generated once).
Each entry: env assignments separated by spaces."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sim3opt_amd import lib as L, synth
synth.DRIFT_TARGET = 0.05
g = synth.manhattan(100000, 1000000)
KNOBS = ["SIM3OPT_AMG_OMEGA", "SIM3OPT_AMG_CYCLE", "SIM3OPT_AMG_PASSES", "SIM3OPT_AMG_ADDITIVE", "SIM3OPT_AMG_OVER",
         "SIM3OPT_AMG_FP32", "SIM3OPT_AMG_COARSEST"]
configs = sys.argv[1:] or [""]
for cfg in configs:
    for k in KNOBS: os.environ.pop(k, None)
    for kv in cfg.split():
        k, v = kv.split("="); os.environ["SIM3OPT_AMG_" + k] = v
    G = L.Graph(fix_small_angle_b=1, pcg_rel_tol=1e-8, time_kernels=1)
    G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"]); G.initialize()
    G.optimize(2); G.set_vertices(g["states"])
    t = time.perf_counter(); G.optimize(8); dt = time.perf_counter() - t
    st = G.stats(); its = [s.pcg_iters for s in st]
    print("%-40s levels %s  %.2f LM it/s  chi %.6g  pcg %s  ms/pcg-it %.3f" % (
        cfg, list(G.amg_hierarchy()[0]), 8 / dt, st[-1].chi2_after, its, sum(s.ms_solve for s in st) / max(1, sum(its))), flush=True)
    G.close()
