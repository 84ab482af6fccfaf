#!/bin/bash
# rocprofv3 --pmc passes (one counter group per run, kernel trace only) for the PCG's SpMV kernel on
# the config-3 matrix; per-launch means go to gpurun_out/pmc_spmv.json.  Run on the GPU box.
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
# TCC has 4 counter slots per pass: FETCH_SIZE costs 3, WRITE_SIZE 2 (MI355X_MICROARCH.md, PMC slots)
for grp in "FETCH_SIZE" "WRITE_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  echo "pass $i: $grp"
  REPS=10 timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o pmc -- python3 $REPO/scripts/gpu_spmv_one.py > $OUT/p$i.log 2>&1
done
cd $REPO && python3 scripts/pmc_reduce.py $OUT gpurun_out/pmc_spmv.json
