"""Two-phase upper-triangle SpMV prototype against the product SpMV on the config-3 matrix.
Variant 0: round 2's naive phase 1 (a wavefront per row); variant 1 (round 3): row spans, software
pipelining, shared gather, batched t stores; SIM3OPT_SPAN_GRID sweeps its grid."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sim3opt_amd import lib as L, synth
synth.DRIFT_TARGET = 0.05
g = synth.manhattan()
G = L.Graph(fix_small_angle_b=1, preconditioner=0)
G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"]); G.initialize()
G.linearize()
nb, nnzb = G.system_dims()
full = nnzb * 396 + (nb + 1) * 4 + 2 * 7 * nb * 8
for variant, grid in ((0, None), (1, None), (1, 3072), (1, 4096), (1, 8192), (1, 12288), (0, None), (1, None)):
    os.environ["SIM3OPT_SYMM_VARIANT"] = str(variant)
    if grid: os.environ["SIM3OPT_SPAN_GRID"] = str(grid)
    else: os.environ.pop("SIM3OPT_SPAN_GRID", None)
    ms = G.bench_spmv(30)
    p1, p2, err, byt = G.bench_spmv_symmetric(30)
    print("variant %d grid %s: full SpMV %.4f ms (%.0f MB, %.0f GB/s) | symmetric: phase 1 %.4f ms + phase 2 %.4f ms = %.4f ms (%.0f MB, %.0f GB/s), "
          "max rel diff %.2e" % (variant, grid, ms, full / 1e6, full / ms / 1e6, p1, p2, p1 + p2, byt / 1e6, byt / (p1 + p2) / 1e6, err), flush=True)
