#!/bin/bash
# The round's final measurement set (run on the MI355X box; copy gpurun_out/final/* into profiles/ as r<N>_final_*):
#   1. the driver's command, unprofiled            -> final/bench_steps20.json
#   2. the default window, unprofiled              -> final/bench.json
#   3. the driver's command's timed path under rocprofv3 --kernel-trace --stats
#                                                  -> final/kernel_stats.csv, final/under_rocprof.json,
#                                                     final/spmv_working_dispatches.json
mkdir -p gpurun_out/final
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final/bench_steps20.json 2> gpurun_out/final/bench_steps20.err || exit 1
python3 bench.py --main-only --no-cpu-baseline > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err || exit 1
bash scripts/prof_bench.sh final --steps 20 --warmup 5 > gpurun_out/final/prof.txt 2>&1 || exit 1
cp gpurun_out/prof_bench/final_kernel_stats.csv gpurun_out/final/kernel_stats.csv
cp gpurun_out/prof_bench_final_line.json gpurun_out/final/under_rocprof.json
python3 scripts/rocprof_filtered_avg.py gpurun_out/prof_bench/final_kernel_trace.csv gpurun_out/final/spmv_working_dispatches.json
head -30 gpurun_out/final/prof.txt
