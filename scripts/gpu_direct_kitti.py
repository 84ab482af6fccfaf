"""Exploration (GPU box): the exact block Cholesky on the KITTI-00 graphs in the reference's own
configuration (delta = 1e-9, B as written, 100 LM iterations) next to the CPU oracle."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle as O
from sim3opt_amd import lib as L, synth
import kitti_graph as K

def dense_from_system(G):
    return G.dense_system()

out = {}
for name, one in (("one_loop", True), ("all_118_loops", False)):
    g = K.build_direct_graph(one)
    G = L.Graph(verbose=int(os.environ.get("VERBOSE", "0")), time_kernels=1)  # (time_kernels: per-phase times on a small graph)
    G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"]); G.initialize()
    print(name, "linear solver in use:", G.linear_solver_in_use(), flush=True)
    G.linearize()
    M, b = dense_from_system(G)
    for lam in (1e-3, 1.0, 1e4):
        x, it, rr = G.solve(lam)
        xr = np.linalg.solve(M + lam * np.eye(M.shape[0]), b)
        print("  lambda %g: |x - dense| / |x| = %.3e   residual %.3e" % (
            lam, np.abs(x - xr).max() / np.abs(xr).max(),
            np.abs((M + lam * np.eye(M.shape[0])) @ x - b).max() / np.abs(b).max()), flush=True)
    # timing of the bare solve
    t = time.perf_counter()
    for _ in range(50): G.solve(1.0)
    print("  solve (incl. copy back + sync): %.1f us" % (1e6 * (time.perf_counter() - t) / 50), flush=True)
    G.set_vertices(g["states"])
    t = time.perf_counter(); n = G.optimize(100); dt = time.perf_counter() - t
    st = G.stats()
    G.set_vertices(g["states"])
    t = time.perf_counter(); n2 = G.optimize(100); dt2 = time.perf_counter() - t
    OG = O.Graph(g["states"], g["fixed"], g["v0"], g["v1"], g["meas"])
    t = time.perf_counter(); it, tr = OG.optimize(100, O.default_options()); dtc = time.perf_counter() - t
    rm = synth.rmse(G.get_vertices(), OG.states)
    rec = dict(gpu_iters=n, gpu_s=dt, gpu_s_second_run=dt2, gpu_chi2=st[-1].chi2_after, oracle_iters=it, oracle_s=dtc,
               oracle_chi2=tr[-1].chi2_after, rmse_gpu_vs_oracle=rm,
               gpu_trials=[s.trials for s in st[:20]], oracle_trials=[r.trials for r in tr[:20]],
               gpu_chi2_head=[s.chi2_after for s in st[:12]], oracle_chi2_head=[r.chi2_after for r in tr[:12]],
               gpu_ms=dict(lin=sum(s.ms_linearize for s in st), solve=sum(s.ms_solve for s in st), upd=sum(s.ms_update for s in st)))
    print(json.dumps(rec), flush=True)
    out[name] = rec
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "direct_kitti.json"), "w"), indent=1)
