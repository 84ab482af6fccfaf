#!/bin/bash
# rocprofv3 --kernel-trace --stats of the timed path of bench.py -> gpurun_out/prof_bench/<tag>_kernel_stats.csv
TAG=${1:-bench}
shift
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -o $TAG -- python3 bench.py --main-only --no-cpu-baseline "$@" > gpurun_out/prof_bench_$TAG.log 2>&1
grep -m1 "^{\"metric\"" gpurun_out/prof_bench_$TAG.log > gpurun_out/prof_bench_${TAG}_line.json
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_bench/${TAG}_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms %.1f"%(tot/1e6))
for r in rows[:28]:
    print("%8.2f ms %6d x %8.1f us  %s"%(float(r["TotalDurationNs"])/1e6,int(r["Calls"]),float(r["AverageNs"])/1e3,r["Name"][:110]))
PY
