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
# one GPU with ITS best cycle (2/3/3): ms per PCG iteration and iterations per solve, from the single-GPU profiles
one_gpu = {"": dict(ms_it=0.74, its=[22, 23, 26, 31], src="profiles/r4_final_bench_steps20.json (LM iterations 1-4)"),
           "_V1000000": dict(ms_it=5.43, its=[33, 10, 6, 6, 8, 11], src="profiles/r3_scale_1M_10M.json")}
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
        t1 = rep["ms_solve_per_pcg_iteration"]
        trep = rep["ms_replicated_levels_per_pcg_iteration"]
        tN = (t1 - trep) / N + trep + ncoll * LAT * 1e-3 + mb / BW  # MB / (GB/s) = ms
        its_N = rep["pcg_iters"][:len(one_gpu[tag]["its"])]
        ratio_its = sum(its_N) / sum(one_gpu[tag]["its"])
        speed = one_gpu[tag]["ms_it"] / (tN * ratio_its)
        print(f"{name}, N = {N}: levels {rep['levels_rows']}; one GPU running the {N}-rank hierarchy {t1:.3f} ms per PCG iteration, "
              f"replicated levels {trep:.3f} ms ({100 * trep / t1:.1f} %)")
        print(f"   partitioned part / {N} = {(t1 - trep) / N:.3f} ms + replicated {trep:.3f} + {ncoll:.1f} collectives x "
              f"{LAT:.0f} us = {ncoll * LAT * 1e-3:.3f} + {mb:.2f} MB / {BW:.0f} GB/s = {mb / BW:.3f}  ->  {tN:.3f} ms per iteration")
        print(f"   one GPU, its own cycle: {one_gpu[tag]['ms_it']:.2f} ms per iteration ({one_gpu[tag]['src']}); PCG iterations "
              f"{sum(its_N)} against {sum(one_gpu[tag]['its'])} (x{ratio_its:.2f})  ->  PCG part {speed:.2f}x at N = {N}")
