#!/bin/bash
# SQ / TA / TCC counters of the level-0 passes of the BATCHED solve (k_spmv_span<..., K = 4>) next to their
# one-system twins, per-launch means -> gpurun_out/pmc_sq_batch.json.  Run on the GPU box (from the repo root).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmcsqb
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_WAVES" \
           "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES" "WRITE_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  echo "pass $i: $grp"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o pmc -- python3 $REPO/scripts/gpu_batch_one.py > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
cd $REPO && python3 - <<'PY'
import csv, glob, json
names = {"k_spmv_span<8, true, 0, double, 1, false>": "one_system_pcg_spmv_fp64", "k_spmv_span<8, true, 1, float, 1, false>": "one_system_residual_fp32",
         "k_spmv_span<8, true, 2, float, 1, false>": "one_system_smoothing_fp32",
         "k_spmv_span<8, true, 0, double, 4, false>": "four_systems_pcg_spmv_fp64",
         "k_spmv_span<8, true, 1, float, 4, false>": "four_systems_residual_fp32",
         "k_spmv_span<8, true, 2, float, 4, false>": "four_systems_smoothing_fp32"}
acc = {}
for f in glob.glob("gpurun_out/pmcsqb/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k, tag in names.items():
            if k in r["Kernel_Name"]:
                a = acc.setdefault(tag, {}).setdefault(r["Counter_Name"], [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
out = {tag: {c: v[1] / v[0] for c, v in d.items()} for tag, d in acc.items()}
for tag, d in out.items():
    d["dispatches"] = max(v[0] for v in acc[tag].values())
    if "TCC_EA0_RDREQ_sum" in d:
        d["hbm_read_bytes"] = d["TCC_EA0_RDREQ_sum"] * 128.0   # gfx950: 128-B requests (MI355X_MICROARCH.md)
        d["hbm_write_bytes"] = d.get("WRITE_SIZE", 0.0) * 1024.0
json.dump(out, open("gpurun_out/pmc_sq_batch.json", "w"), indent=1)
for tag in sorted(out):
    d = out[tag]
    print(tag, {k: (round(v) if v > 100 else round(v, 3)) for k, v in d.items()})
PY
