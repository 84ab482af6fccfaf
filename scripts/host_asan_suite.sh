#!/bin/bash
# The whole library with its HOST code under AddressSanitizer + UndefinedBehaviorSanitizer (hipcc instruments the host side;
# the gfx950 code objects are unchanged -- GPU sanitizers do not exist on this pool), then every non-GPU test that drives
# the library through its C-ABI: loaders of the reference's file formats, patterns, partitions, plans, hierarchy, the
# gloo emulation of the partitioned path.  Needs no GPU.  A sanitizer report aborts the run.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/libsim3opt_asan.so}
RT=$(find /opt/rocm/lib/llvm/lib/clang -name "libclang_rt.asan-x86_64.so" | head -1)
cd "$ROOT/sim3opt_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined \
    -fno-sanitize-recover=all -Wno-unused-result -Wno-option-ignored -o "$OUT" $(cat SOURCES) -ldl
cd "$ROOT"
SIM3OPT_LIB="$OUT" LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    python -m pytest tests/test_host.py tests/test_formats.py tests/test_direct_plan.py tests/test_ba.py \
    tests/test_reference_pins.py tests/test_distributed_cpu.py -q -m "not gpu" -p no:cacheprovider
