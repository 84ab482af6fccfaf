"""Does a locality-preserving vertex order speed up the SpMV?  (tuning experiment)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sim3opt_amd import lib as L, synth
synth.DRIFT_TARGET = 0.05
g = synth.manhattan()
V = g['states'].shape[0]
def run(tag, perm):
    inv = np.empty(V, dtype=np.int64); inv[perm] = np.arange(V)   # new index of old vertex
    G = L.Graph(fix_small_angle_b=1)
    G.add_vertices(g['states'][perm], g['fixed'][perm])
    G.add_edges(inv[g['v0']], inv[g['v1']], g['meas'])
    G.initialize(); G.linearize()
    nb, nnzb = G.system_dims()
    ms = min(G.bench_spmv(50) for _ in range(3))
    print(tag, "ms %.4f" % ms, "GB/s %.0f" % ((nnzb * 396 + (nb + 1) * 4 + 2 * 7 * nb * 8) / ms / 1e6), flush=True)
    G.close()
pos = np.rint(synth.positions(g['gt'])).astype(np.int64)
pos -= pos.min(0)
run("walk order", np.arange(V))
cell = (pos[:, 0] * 100 + pos[:, 1]) * 16 + pos[:, 2]
p = np.argsort(cell, kind="stable"); p = np.concatenate([[0], p[p != 0]])
run("cell (x,y,z) order", p)
def morton(a):
    out = np.zeros(len(a), dtype=np.int64)
    for b in range(8):
        for d in range(3):
            out |= ((a[:, d] >> b) & 1) << (3 * b + d)
    return out
p = np.argsort(morton(pos), kind="stable"); p = np.concatenate([[0], p[p != 0]])
run("morton order", p)
rng = np.random.default_rng(0); p = rng.permutation(V); p = np.concatenate([[0], p[p != 0]])
run("random order", p)
