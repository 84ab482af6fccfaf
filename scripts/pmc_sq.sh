#!/bin/bash
# SQ / TA counters of the PCG's SpMV kernel (where do the waves wait?): per-launch means -> gpurun_out/pmc_sq.json
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmcsq
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_WAVES" \
           "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES" "TA_DATA_STALLED_BY_TC_CYCLES TA_ADDR_STALLED_BY_TD_CYCLES"; do
  i=$((i+1))
  echo "pass $i: $grp"
  REPS=10 timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o pmc -- python3 $REPO/scripts/gpu_spmv_one.py > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
cd $REPO && python3 - <<'PY'
import csv, glob, json, os
acc = {}
for f in glob.glob("gpurun_out/pmcsq/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_spmv_span<8, true, 0, double, 1, false>" not in r["Kernel_Name"]:
            continue
        a = acc.setdefault(r["Counter_Name"], [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
out = {k: v[1] / v[0] for k, v in acc.items()}
json.dump(out, open("gpurun_out/pmc_sq.json", "w"), indent=1)
print(json.dumps(out))
PY
