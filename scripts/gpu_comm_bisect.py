"""Bisect: does the multi-rank code path (one rank, SIM3OPT_FORCE_COMM=1) change PCG iteration counts on config 3?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sim3opt_amd import lib as L, synth
synth.DRIFT_TARGET = 0.05
g = synth.manhattan()
for force in (0, 1):
    if force: os.environ["SIM3OPT_FORCE_COMM"] = "1"
    G = L.Graph(fix_small_angle_b=1, pcg_rel_tol=1e-8)
    G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"])
    if force:
        uid = np.zeros(128, dtype=np.uint8)
        assert L.load().sim3opt_comm_unique_id(uid.ctypes.data_as(L._up)) == L.OK
        G.comm_init_rccl(0, 1, uid)
    G.initialize(); G.optimize(4)
    st = G.stats()
    print("force_comm", force, [s.pcg_iters for s in st], ["%.10g" % s.chi2_after for s in st], ["%.2e" % s.pcg_rel_res for s in st], flush=True)
    G.close()
