"""Row-per-lane prototype of the cycle's level-0 FP32 passes (csrc/rowlane_proto.hpp) against the product kernel on
the config-3 matrix.  Needs a SIM3OPT_BENCH_HOOKS build: SIM3OPT_BENCH_HOOKS=1 python -m sim3opt_amd.build --force"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sim3opt_amd import lib as L, synth
synth.DRIFT_TARGET = 0.05
g = synth.manhattan()
G = L.Graph(fix_small_angle_b=1)
G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"]); G.initialize()
assert G.optimize(2) == 2
nb, nnzb = G.system_dims()
mb32 = (nnzb * 200 + (nb + 1) * 4 + 3 * 7 * nb * 8) / 1e6
for rows in [int(a) for a in sys.argv[1:]] or [12500, 18000, 25000, 37500, 50000, 100000]:
    o = G.bench_spmv_rowlane(30, rows)
    print("rows per group %d: product residual %.4f / smoothing %.4f ms | prototype %.4f / %.4f ms (%.0f / %.0f GB/s) | four systems %.4f / %.4f ms "
          "| max rel diff %.1e / %.1e" % (rows, o[0], o[1], o[2], o[3], mb32 / o[2], (mb32 + 49 * nb * 8 / 1e6) / o[3], o[4], o[5], o[6], o[7]), flush=True)
