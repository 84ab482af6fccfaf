"""BASELINE.json configs[1]: synthetic chain + loop graph, 10k vertices / 20k edges, 10 LM iterations on
the GPU.  The CPU oracle cannot factor this graph (10 001 random long-range loops: an expander, the
fill of any elimination order is near-dense), so only its linearisation + chi2 phases are timed -- an
upper bound on its LM rate, as in bench.py's cpu_baseline.  Writes gpurun_out/r4_config2.json."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O
from sim3opt_amd import lib as L, synth
synth.DRIFT_TARGET = 0.05
g = synth.chain_loop(10000, 20000)
out = {}
for fixb in (1, 0):
    G = L.Graph(fix_small_angle_b=fixb, pcg_rel_tol=1e-8)
    G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"]); G.initialize()
    G.optimize(1); G.set_vertices(g["states"])
    chi0 = G.chi2()
    t = time.perf_counter(); n = G.optimize(10); t_gpu = time.perf_counter() - t
    st = G.stats()
    OG = O.Graph(g["states"], g["fixed"], g["v0"], g["v1"], g["meas"])
    o = O.default_options(fix_small_angle_b=fixb)
    t = time.perf_counter(); OG.chi2(o); OG.jacobians(o); t_cpu = time.perf_counter() - t
    out[f"fix_small_angle_b={fixb}"] = dict(
        gpu_iters=n, gpu_seconds=t_gpu, gpu_lm_iters_per_s=n / t_gpu,
        solver="exact" if G.linear_solver_in_use() else "pcg, preconditioner %d" % G.preconditioner_in_use(),
        chi2_initial=chi0, gpu_chi2=[s.chi2_after for s in st], gpu_trials=[s.trials for s in st],
        gpu_pcg_iters=[s.pcg_iters for s in st], gpu_pcg_rel_res=[s.pcg_rel_res for s in st],
        cpu_seconds_linearize_plus_chi2=t_cpu, cpu_cores=1, speedup_lower_bound=t_cpu / (t_gpu / max(n, 1)))
    print(json.dumps(out[f"fix_small_angle_b={fixb}"]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r4_config2.json"), "w"), indent=1)
