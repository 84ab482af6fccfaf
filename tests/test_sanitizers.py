"""AddressSanitizer + UndefinedBehaviorSanitizer on the CPU code (not gpu): GPU sanitizers do not exist on this
pool, and the index arithmetic that feeds the kernels -- pattern, partitions, halo plans, the
partition-aware hierarchy, the elimination plan -- is host code; so is the oracle every parity test trusts.
Each driver checks its invariants and ends with "ok"; a sanitizer report ends it with an error."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sim3opt_amd", "csrc")
SAN = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")


def _have(cc):
    if shutil.which(cc) is None:
        return False
    r = subprocess.run([cc, "-fsanitize=address,undefined", "-x", "c", "-", "-o", os.devnull], input="int main(void){return 0;}",
                       capture_output=True, text=True)
    return r.returncode == 0


def _run(exe):
    r = subprocess.run([exe], capture_output=True, text=True, env=ENV, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    assert r.stdout.strip().endswith("ok"), r.stdout[-2000:]
    return r.stdout


@pytest.mark.skipif(not _have("gcc"), reason="gcc with the sanitizer runtimes is not installed")
def test_oracle_under_asan_ubsan(tmp_path):
    """oracle/sim3_oracle.c through every entry point of its header: per-edge residuals and numeric Jacobians, the
    dense system, one solve, LM with and without information matrices / Huber kernel in both arithmetics, the
    nothing-to-optimise return."""
    exe = str(tmp_path / "oracle_san")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-D_POSIX_C_SOURCE=200809L"] + SAN +
                          ["-I" + os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "cxx", "oracle_sanitizer_driver.c"),
                           os.path.join(ROOT, "oracle", "sim3_oracle.c"), "-lm", "-o", exe])
    out = _run(exe)
    assert out.count("iterations, chi2") == 3 and "all fixed: -1" in out


@pytest.mark.skipif(not _have("g++") or not os.path.exists("/opt/rocm/include/hip/hip_runtime.h"),
                    reason="g++ with the sanitizer runtimes (and the HIP headers sim3_math.hpp includes) not installed")
def test_host_algorithms_under_asan_ubsan(tmp_path):
    """csrc/graph.cpp, amg.cpp, direct.cpp (pure host C++): block-CSR pattern in insertion and locality order with
    arbitrary ids, parallel edges and two fixed vertices; row partitions and neighbour halo plans for 1 ... 64 ranks
    (what p sends to q is what q receives from p); the aggregation hierarchy for 1 / 2 / 4 / 8 ranks and three
    partition thresholds (aggregates inside the ranks' spans, contiguous coarse spans); elimination plans under three
    schedules and the refused plan; malformed graphs."""
    exe = str(tmp_path / "host_san")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + CSRC] + SAN +
                          [os.path.join(ROOT, "tests", "cxx", "host_sanitizer_driver.cpp")] +
                          [os.path.join(CSRC, f) for f in ("graph.cpp", "amg.cpp", "direct.cpp")] + ["-o", exe])
    out = _run(exe)
    assert "hierarchies" in out
