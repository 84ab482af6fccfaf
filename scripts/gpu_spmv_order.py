"""Does a locality-preserving vertex order speed the SpMV up?  The config-3 graph as generated (vertex
order = order along the random walk) against the same graph relabelled by lattice position
(lexicographic / Morton), SpMV back to back in one process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sim3opt_amd import lib as L, synth
synth.DRIFT_TARGET = 0.05
g = synth.manhattan()
pos = np.round(g["gt"][:, 4:7]).astype(np.int64)
pos -= pos.min(axis=0)


def morton(p):
    out = np.zeros(len(p), dtype=np.int64)
    for b in range(10):
        for d in range(3):
            out |= ((p[:, d] >> b) & 1) << (3 * b + d)
    return out


orders = {"as generated": np.arange(len(pos)),
          "lexicographic (x, y, z)": np.lexsort((pos[:, 2], pos[:, 1], pos[:, 0])),
          "morton": np.argsort(morton(pos), kind="stable")}
for name, order in orders.items():
    new_id = np.empty(len(order), dtype=np.int64)
    new_id[order] = np.arange(len(order))
    G = L.Graph(fix_small_angle_b=1, preconditioner=0)
    G.add_vertices(g["states"][order], g["fixed"][order])
    G.add_edges(new_id[g["v0"]], new_id[g["v1"]], g["meas"])
    G.initialize(); G.linearize()
    nb, nnzb = G.system_dims()
    ms = [G.bench_spmv(30) for _ in range(3)]
    byt = nnzb * 396 + (nb + 1) * 4 + 2 * 7 * nb * 8
    print("%-26s SpMV %.4f / %.4f / %.4f ms  (%.0f GB/s best)" % (name, ms[0], ms[1], ms[2], byt / min(ms) / 1e6), flush=True)
    G.close()
