"""Round-3 sweep of the multigrid knobs on config 3 (one process, graph generated once): Chebyshev
smoother degrees per level (CHEB), their interval (CHEB_B, CHEB_ALPHA), cycle shape, aggregate sizes.
Each argument: knob assignments separated by spaces, without the SIM3OPT_AMG_ prefix.
NIT / NWARM: LM iterations timed / warm-up (default 8 / 2)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sim3opt_amd import lib as L, synth
synth.DRIFT_TARGET = 0.05
g = synth.manhattan(100000, 1000000)
NIT = int(os.environ.get("NIT", "8")); NWARM = int(os.environ.get("NWARM", "2"))
configs = sys.argv[1:] or [""]
for cfg in configs:
    for k in list(os.environ):
        if k.startswith("SIM3OPT_AMG_"): os.environ.pop(k)
    for kv in cfg.split():
        k, v = kv.split("="); os.environ["SIM3OPT_AMG_" + k] = v
    try:
        G = L.Graph(fix_small_angle_b=1, pcg_rel_tol=1e-8, time_kernels=1, preconditioner=int(os.environ.get("PREC", "-1")))
        G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"]); G.initialize()
        G.optimize(NWARM); G.set_vertices(g["states"])
        t = time.perf_counter(); G.optimize(NIT); dt = time.perf_counter() - t
        st = G.stats(); its = [s.pcg_iters for s in st]
        print("%-44s levels %s  %.2f LM it/s  chi %.6g  pcg %s = %d  ms/pcg-it %.3f  trials %d" % (
            cfg, [int(x) for x in G.amg_hierarchy()[0]], NIT / dt, st[-1].chi2_after, its, sum(its),
            sum(s.ms_solve for s in st) / max(1, sum(its)), sum(s.trials for s in st)), flush=True)
        G.close()
    except Exception as e:  # a knob combination the library refuses
        print("%-44s FAILED: %s" % (cfg, e), flush=True)
