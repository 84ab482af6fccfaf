"""What one RCCL collective costs at the very least on this box: the RCCL transport with ONE rank (forced collectives:
in-place ncclAllReduce of 16 bytes, ncclAllGather / grouped in-place ncclBroadcast, self-addressed grouped
ncclSend / ncclRecv -- every call of the partitioned path) on the library's stream, device time between HIP events
(sim3opt_get_comm_times).  No link is crossed: this is the launch + completion floor of a collective, the part of
DESIGN.md 7's "25 us per collective" that does not depend on xGMI.  Config 3 (100k / 1M), 3 LM iterations."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SIM3OPT_FORCE_COMM"] = "1"
from sim3opt_amd import lib as L, synth
synth.DRIFT_TARGET = 0.05
g = synth.manhattan()
out = {}
for prec in (0, 2):
    uid = np.zeros(128, dtype=np.uint8)  # (an id serves one communicator)
    assert L.load().sim3opt_comm_unique_id(uid.ctypes.data_as(L._up)) == L.OK
    G = L.Graph(fix_small_angle_b=1, pcg_rel_tol=1e-8, preconditioner=prec, time_kernels=1, pcg_max_iters=60)
    G.add_vertices(g["states"], g["fixed"]); G.add_edges(g["v0"], g["v1"], g["meas"])
    G.comm_init_rccl(0, 1, uid)
    G.initialize()
    G.optimize(1)
    G.kernel_times(reset=True)
    G.optimize(3)
    ct = G.comm_times()
    npcg = sum(s.pcg_iters for s in G.stats()[1:])
    rec = dict(pcg_iterations=int(npcg))
    for k in ("allreduce", "allgather", "exchange"):
        n = max(1, ct["n_" + k])
        rec[k] = dict(calls=int(ct["n_" + k]), us_per_call=1e3 * ct["ms_" + k] / n, bytes_per_call=ct["bytes_" + k] / n)
    out["block_jacobi" if prec == 0 else "multigrid"] = rec
    G.close()
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(os.environ.get("OUT", os.path.join(ROOT, "gpurun_out")), "r4_rccl_floor.json"), "w"), indent=1)
