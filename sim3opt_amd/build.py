"""Builds libsim3opt.so in-tree with hipcc for gfx950 (MI355X).

    python -m sim3opt_amd.build [--force]

The .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsim3opt.so")
# the one list of translation units, shared with the Makefile
SOURCES = open(os.path.join(CSRC, "SOURCES")).read().split()
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".hpp")) + [
    os.path.join("..", "..", "include", "sim3opt.h"), os.path.join("..", "..", "include", "sim3opt_bench.h"), "SOURCES"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-result"]
FLAGS += os.environ.get("SIM3OPT_EXTRA_FLAGS", "").split()  # (A/B experiments build twice in one gpurun call)
# measurement prototypes (engine_proto.hip) are NOT part of the product library
if os.environ.get("SIM3OPT_BENCH_HOOKS", "0") not in ("", "0"):
    FLAGS.append("-DSIM3OPT_BENCH_HOOKS")
    SOURCES.append("engine_proto.hip")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    cmd = [HIPCC] + FLAGS + ["-o", LIB] + srcs + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
