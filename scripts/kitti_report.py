"""KITTI-00 report (BASELINE.json configs[0] and configs[4]): GPU (libsim3opt) next to the CPU
oracle, 100 LM iterations like the reference (kitti_surf.cpp:675), both arithmetics, with the
reference's own quality metric (RMSE vs KITTI ground truth after Umeyama, kitti_surf.cpp:1427-1463).
Writes profiles/r1_kitti_report.json.  Run on the GPU box."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle as O
from sim3opt_amd import lib as L, synth
import kitti_graph as K

gt = np.loadtxt(os.path.join(K.FIXTURE, "gt_kf.txt"), comments="%")[:, [4, 8, 12]]
out = {"original_map_rmse_m": L.align_trajectory(synth.positions(K.build_direct_graph(True)["states"]), gt)[1]}
for name, one in (("one_loop", True), ("all_118_loops", False)):
    g = K.build_direct_graph(one)
    for fixb in (0, 1):
        rec = {}
        G = L.Graph(fix_small_angle_b=fixb, pcg_rel_tol=1e-10, pcg_max_iters=20000)
        G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"]); G.initialize()
        chi0 = G.chi2()
        t = time.perf_counter(); n = 0; st = []
        while n < 100:
            k = G._L.sim3opt_optimize(G._g, 100 - n)
            if k <= 0: break
            n += k; st += G.stats()
            if st[-1].trials >= 10 or st[-1].rho == 0: break   # g2o Terminate
        dt = time.perf_counter() - t
        rec["gpu"] = dict(iters=n, seconds=dt, chi2_0=chi0, chi2_final=st[-1].chi2_after,
                          pcg_iters_total=int(sum(s.pcg_iters for s in st)),
                          rmse_vs_gt_m=L.align_trajectory(synth.positions(G.get_vertices()), gt)[1],
                          scale_range=[float(G.get_vertices()[:, 7].min()), float(G.get_vertices()[:, 7].max())])
        OG = O.Graph(g["states"], g["fixed"], g["v0"], g["v1"], g["meas"])
        t = time.perf_counter(); it, tr = OG.optimize(100, O.default_options(fix_small_angle_b=fixb)); dtc = time.perf_counter() - t
        rec["cpu_oracle"] = dict(iters=it, seconds=dtc, chi2_final=tr[-1].chi2_after,
                                 rmse_vs_gt_m=L.align_trajectory(synth.positions(OG.states), gt)[1])
        rec["rmse_gpu_vs_oracle_m"] = synth.rmse(G.get_vertices(), OG.states)
        out[f"{name}/fix_small_angle_b={fixb}"] = rec
        print(name, fixb, json.dumps(rec), flush=True)
# config 5 (a): the reference's stepwise staging (kitti_surf.cpp:887-1047), 100 + 100 iterations
for fixb in (0, 1):
    g = K.build_direct_graph(False)
    G = L.Graph(fix_small_angle_b=fixb, pcg_rel_tol=1e-10, pcg_max_iters=20000)
    G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"])
    t0 = time.perf_counter(); G.stepwise_scale_init(); t1 = time.perf_counter()
    r1 = L.align_trajectory(synth.positions(G.get_vertices()), gt)[1]
    G.set_options(dof_mask=0x78); G.initialize()
    n2 = 0
    while n2 < 100:
        k = G._L.sim3opt_optimize(G._g, 100 - n2)
        if k <= 0: break
        n2 += k
        if G.stats()[-1].trials >= 10 or G.stats()[-1].rho == 0: break
    t2 = time.perf_counter(); chi2_2 = G.stats()[-1].chi2_after
    r2 = L.align_trajectory(synth.positions(G.get_vertices()), gt)[1]
    G.set_options(dof_mask=127)
    n3 = 0
    while n3 < 100:
        k = G._L.sim3opt_optimize(G._g, 100 - n3)
        if k <= 0: break
        n3 += k
        if G.stats()[-1].trials >= 10 or G.stats()[-1].rho == 0: break
    t3 = time.perf_counter()
    out[f"stepwise_all_loops/fix_small_angle_b={fixb}"] = dict(
        scale_dlt_s=t1 - t0, rmse_after_scales_m=r1, scale_trans_iters=n2, scale_trans_s=t2 - t1,
        scale_trans_chi2=chi2_2, rmse_after_scale_trans_m=r2, sim3_iters=n3, sim3_s=t3 - t2,
        sim3_chi2=G.stats()[-1].chi2_after,
        rmse_vs_gt_m=L.align_trajectory(synth.positions(G.get_vertices()), gt)[1])
    print("stepwise", fixb, json.dumps(out[f"stepwise_all_loops/fix_small_angle_b={fixb}"]), flush=True)
# config 5 (b): incremental loop closures, warm-started LM (<= 20 iterations) per closure
full = K.build_direct_graph(False); nl = 118
G = L.Graph(fix_small_angle_b=1, pcg_rel_tol=1e-10, pcg_max_iters=20000)
G.add_vertices(full["states"], full["fixed"]); G.add_edges(full["v0"][nl:], full["v1"][nl:], full["meas"][nl:]); G.initialize()
per = []
t0 = time.perf_counter()
for k in range(nl):
    G.add_edge(int(full["v0"][k]), int(full["v1"][k]), full["meas"][k]); G.initialize()
    t = time.perf_counter(); n = G.optimize(20); per.append(dict(closure=k, iters=n, ms=1e3 * (time.perf_counter() - t), chi2=G.stats()[-1].chi2_after))
out["incremental_closures_fixb1"] = dict(total_seconds=time.perf_counter() - t0, closures=nl,
    mean_ms_per_closure=float(np.mean([p["ms"] for p in per])), final_chi2=per[-1]["chi2"],
    rmse_vs_gt_m=L.align_trajectory(synth.positions(G.get_vertices()), gt)[1], first=per[:3], last=per[-3:])
print(json.dumps(out["incremental_closures_fixb1"]))
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r1_kitti_report.json"), "w"), indent=1)
