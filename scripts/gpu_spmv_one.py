"""Runs the SpMV kernel a few times on the config-3 matrix (for rocprofv3 --pmc passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sim3opt_amd import lib as L, synth
synth.DRIFT_TARGET = 0.05
g = synth.manhattan()
G = L.Graph(fix_small_angle_b=1); G.add_vertices(g['states'], g['fixed']); G.add_edges(g['v0'], g['v1'], g['meas']); G.initialize()
G.linearize()
nb, nnzb = G.system_dims()
ms = G.bench_spmv(int(os.environ.get("REPS", "10")))
print("ms", ms, "alg bytes", nnzb * 396 + (nb + 1) * 4 + 3 * 7 * nb * 8)
for mode in (0, 1, 2, 0, 1, 2):
    ms = G.bench_stream(mode, 30)
    print("stream mode", mode, "ms %.4f" % ms, "GB/s %.0f" % (nnzb * 392 / ms / 1e6))
