import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sim3opt_amd import lib as L, synth
import kitti_graph as K
for one in (True, False):
    g = K.build_direct_graph(one)
    for passes, omega in (("3", "0.9"), ("2", "0.9"), ("2", "0.7"), ("3", "0.7")):
        os.environ["SIM3OPT_AMG_PASSES"] = passes; os.environ["SIM3OPT_AMG_OMEGA"] = omega
        G = L.Graph(fix_small_angle_b=1, pcg_rel_tol=1e-10, pcg_max_iters=40000)
        G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"]); G.initialize()
        rows = G.amg_hierarchy()[0]
        G.optimize(1); G.set_vertices(g["states"])
        t = time.perf_counter(); n = G.optimize(100); dt = time.perf_counter() - t
        st = G.stats()
        print("one_loop %s passes %s omega %s levels %s: %d LM it %.3fs chi %.8g pcg total %d" % (one, passes, omega, list(rows), n, dt, st[-1].chi2_after, sum(s.pcg_iters for s in st)), flush=True)
        G.close()
