import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sim3opt_amd import lib as L, synth
synth.DRIFT_TARGET = 0.05
g = synth.chain_loop()
for pre, env in ((0, {}), (2, {"SIM3OPT_AMG_ADDITIVE": "0", "SIM3OPT_AMG_CYCLE": "13"}), (2, {"SIM3OPT_AMG_ADDITIVE": "1", "SIM3OPT_AMG_CYCLE": "12"})):
    os.environ.update(env)
    G = L.Graph(fix_small_angle_b=1, preconditioner=pre, pcg_rel_tol=1e-8)
    G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"]); G.initialize()
    G.optimize(1); G.set_vertices(g["states"])
    t = time.perf_counter(); n = G.optimize(12); dt = time.perf_counter() - t
    st = G.stats()
    print("config2 pre %d %s: %d LM it %.3fs chi %.8g pcg %s" % (pre, env, n, dt, st[-1].chi2_after, [s.pcg_iters for s in st]), flush=True)
    G.close()
