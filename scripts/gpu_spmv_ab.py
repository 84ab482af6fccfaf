"""A/B of SpMV tuning knobs on the config-3 matrix (one process, back to back).
Knobs are environment variables read at sim3opt_initialize: SIM3OPT_SPMV="chunk,nt" and
SIM3OPT_SPAN_GRID=<workgroups>; edit `variants` / the variable name below for the knob under test."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sim3opt_amd import lib as L, synth
synth.DRIFT_TARGET = 0.05
g = synth.manhattan()
res = {}
KNOB = os.environ.get("KNOB", "SIM3OPT_SPAN_GRID")
variants = os.environ.get("VARIANTS", "2048,4096,6250,8192").split(";") if KNOB != "SIM3OPT_SPAN_GRID" else os.environ.get("VARIANTS", "2048,4096,6250,8192").split(",")
for rep in range(2):
    for v in variants:
        os.environ[KNOB] = v
        G = L.Graph(fix_small_angle_b=1, preconditioner=0); G.add_vertices(g['states'], g['fixed']); G.add_edges(g['v0'], g['v1'], g['meas']); G.initialize()
        G.linearize()
        nb, nnzb = G.system_dims()
        ms = G.bench_spmv(50)
        byt = nnzb * 396 + (nb + 1) * 4 + 2 * 7 * nb * 8
        res.setdefault(v, []).append(ms)
        print(rep, v, "ms %.4f" % ms, "GB/s %.0f" % (byt / ms / 1e6), flush=True)
        G.close()
