"""One LM iteration of config 3 with the multigrid preconditioner (for rocprofv3 --pmc passes over the
level-0 passes of the cycle: k_spmv_span<8, true, {1,2}, float> next to the PCG's <8, true, 0, double>)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sim3opt_amd import lib as L, synth
synth.DRIFT_TARGET = 0.05
g = synth.manhattan()
G = L.Graph(fix_small_angle_b=1, pcg_rel_tol=1e-8, preconditioner=2)
G.add_vertices(g['states'], g['fixed']); G.add_edges(g['v0'], g['v1'], g['meas']); G.initialize()
print("iterations", G.optimize(int(os.environ.get("NIT", "1"))), "pcg", [s.pcg_iters for s in G.stats()])
