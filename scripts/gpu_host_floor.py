"""Host cost of a multigrid PCG iteration issued launch by launch (what a partitioned run does: no graph capture
around collectives): a graph small enough that the GPU work is negligible (3000 / 30 000, three levels), wall time
per PCG iteration with options.pcg_graph = 0 against the captured loop -- the floor under DESIGN.md 7's model."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sim3opt_amd import lib as L, synth
synth.DRIFT_TARGET = 0.05
out = {}
for V, E, dims in ((3000, 30000, (15, 15, 14)), (100000, 1000000, None)):
    g = synth.manhattan(V, E, dims=dims) if dims else synth.manhattan(V, E)
    if dims: os.environ["SIM3OPT_AMG_COARSEST"] = "16"
    else: os.environ.pop("SIM3OPT_AMG_COARSEST", None)
    for graph in (1, 0):
        for cyc in ([0, 0, 0, 0], [1, 2, 2, 2]):
            G = L.Graph(fix_small_angle_b=1, pcg_rel_tol=1e-10, preconditioner=2, pcg_graph=graph, amg_cycle=cyc, adaptive_prec=0)
            G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"]); G.initialize()
            G.optimize(2)
            t = time.perf_counter(); n = G.optimize(6); dt = time.perf_counter() - t
            st = G.stats()[2:]
            its = sum(s.pcg_iters for s in st); ms = sum(s.ms_solve for s in st)
            key = "%d/%d graph=%d cycle=%s" % (V, E, graph, "".join(map(str, cyc[:3])))
            out[key] = dict(levels=G.amg_in_use()["levels"], pcg_iterations=int(its), ms_solve_per_pcg_iteration=ms / max(1, its),
                            wall_ms_per_pcg_iteration=1e3 * dt / max(1, its), lm_iterations=int(n))
            print(key, out[key], flush=True)
            G.close()
json.dump(out, open(os.path.join(os.environ.get("OUT", os.path.join(ROOT, "gpurun_out")), "r4_host_floor.json"), "w"), indent=1)
