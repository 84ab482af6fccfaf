#!/bin/bash
# SQ / TA / TCC counters of the row-per-lane prototype's residual pass next to the product's (per-launch means).
REPO=${GRAFT_REPO_ROOT:-/root/repo}; HERE=$(pwd)
OUT=$REPO/gpurun_out/pmcrl
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_WAVES" \
           "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES" "TA_DATA_STALLED_BY_TC_CYCLES TA_ADDR_STALLED_BY_TD_CYCLES" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  echo "pass $i: $grp"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o pmc -- python3 $HERE/scripts/gpu_spmv_rowlane.py 4 > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
cd $REPO && python3 - <<'PY'
import csv, glob, json
names = {"k_spmv_span<8, true, 1, float, 1, false>": "product_residual", "k_spmv_rowlane<1, 1>": "rowlane_residual", "k_spmv_rowlane<1, 4>": "rowlane_residual_x4"}
acc = {}
for f in glob.glob("gpurun_out/pmcrl/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k, tag in names.items():
            if k in r["Kernel_Name"]:
                a = acc.setdefault(tag, {}).setdefault(r["Counter_Name"], [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
out = {tag: {c: v[1] / v[0] for c, v in d.items()} for tag, d in acc.items()}
keys = sorted({k for v in out.values() for k in v})
print("%-36s" % "" + "".join("%22s" % t for t in out))
for k in keys:
    print("%-36s" % k + "".join("%22.4g" % out[t].get(k, float("nan")) for t in out))
json.dump(out, open("gpurun_out/pmc_rowlane.json", "w"), indent=1)
PY
