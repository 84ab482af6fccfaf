"""DESIGN.md 7's scaling model, re-derived from the measured parts (no GPU needed), for N = 2, 4, 8:
  profiles/r4_replicated_share_N<N>[_V1000000].json  one GPU running the hierarchy and cycle an N-rank partition builds:
                                                     ms per PCG iteration, of which on the replicated levels
  profiles/r4_dryrun_threads_N<N>[_V1000000].json    N thread-ranks: collectives per PCG iteration and their payloads
and two assumptions: LAT microseconds per collective, BW GB/s per rank for its payload.
Usage: python scripts/model_scaling.py [LAT=25] [BW=50]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAT = float(sys.argv[1]) if len(sys.argv) > 1 else 25.0
BW = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
# one GPU with ITS best cycle (2/3/3): solve time per PCG iteration (set-up of the solve included, as in the N-rank
# figure) and iterations per solve, read from the single-GPU profiles of the end of the round
def _one_gpu(path, a, b, line):
    txt = open(os.path.join(ROOT, "profiles", path)).read().strip()
    d = json.loads(txt.splitlines()[-1]) if line else json.loads(txt)
    return dict(ms_it=sum(d["ms_solve"][a:b]) / sum(d["pcg_iters"][a:b]), its=[int(x) for x in d["pcg_iters"][a:b]], a=a, b=b,
                src="profiles/%s (LM iterations %d-%d)" % (path, a + 1, b))
# (1M / 10M: LM iterations 2-6 on both sides -- the first solve of a fresh graph carries the abandoned block-Jacobi probe)
one_gpu = {"": _one_gpu("r4_end_bench_steps20.json", 0, 4, True), "_V1000000": _one_gpu("r4_scale_1M_10M.json", 1, 6, False)}
for tag, name in (("", "100k / 1M"), ("_V1000000", "1M / 10M")):
    for N in (2, 4, 8):
        fr = os.path.join(ROOT, "profiles", "r4_replicated_share_N%d%s.json" % (N, tag))
        fd = os.path.join(ROOT, "profiles", "r4_dryrun_threads_N%d%s.json" % (N, tag))
        if not (os.path.exists(fr) and os.path.exists(fd)):
            continue
        rep, dry = json.load(open(fr)), json.load(open(fd))
        worst = max(dry["per_rank"], key=lambda p: p["exchange_MB_per_pcg_iteration"])
        ncoll = worst["exchanges_per_pcg_iteration"] + worst["allgathers_per_pcg_iteration"] + worst["allreduces_per_pcg_iteration"]
        mb = worst["exchange_MB_per_pcg_iteration"] + worst["allgather_MB_per_pcg_iteration"]
        a, b = one_gpu[tag]["a"], one_gpu[tag]["b"]   # the same LM iterations on both sides (set-up of the solves included)
        t1 = sum(rep["ms_solve"][a:b]) / sum(rep["pcg_iters"][a:b])
        trep = t1 * rep["replicated_share_of_solve"]
        # ... plus what a solve's set-up replicates and the events around the cycle's visits do not see: the dense inverse of
        # the coarsest level, 7 n / 14 launches of 11.3 us (profiles/r4_end_bench_kernel_stats.csv) + fill, once per solve
        dense_ms = (7 * rep["levels_rows"][-1] / 14.0) * 11.3e-3 + 0.05
        trep += dense_ms * (b - a) / sum(rep["pcg_iters"][a:b])
        tN = (t1 - trep) / N + trep + ncoll * LAT * 1e-3 + mb / BW  # MB / (GB/s) = ms
        its_N = rep["pcg_iters"][one_gpu[tag]["a"]:one_gpu[tag]["b"]]
        ratio_its = sum(its_N) / sum(one_gpu[tag]["its"])
        speed = one_gpu[tag]["ms_it"] / (tN * ratio_its)
        print(f"{name}, N = {N}: levels {rep['levels_rows']}; one GPU running the {N}-rank hierarchy {t1:.3f} ms per PCG iteration, "
              f"replicated (coarse levels + dense inverse) {trep:.3f} ms ({100 * trep / t1:.1f} %)")
        print(f"   partitioned part / {N} = {(t1 - trep) / N:.3f} ms + replicated {trep:.3f} + {ncoll:.1f} collectives x "
              f"{LAT:.0f} us = {ncoll * LAT * 1e-3:.3f} + {mb:.2f} MB / {BW:.0f} GB/s = {mb / BW:.3f}  ->  {tN:.3f} ms per iteration")
        print(f"   one GPU, its own cycle: {one_gpu[tag]['ms_it']:.2f} ms per iteration ({one_gpu[tag]['src']}); PCG iterations "
              f"{sum(its_N)} against {sum(one_gpu[tag]['its'])} (x{ratio_its:.2f})  ->  PCG part {speed:.2f}x at N = {N}")
