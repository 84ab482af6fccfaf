"""The reference-side binding (include/sim3opt_g2o.hpp) in action: examples/direct_pgo.cpp is
testDirectSim3Optimization (kitti_surf.cpp:542-709) on the g2o-named shim."""
import os
import subprocess

import numpy as np
import pytest

import kitti_graph as K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compile_example(tmp_path):
    exe = str(tmp_path / "direct_pgo")
    libdir = os.path.join(ROOT, "sim3opt_amd")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-DSIM3OPT_G2O_NAMES",
                           "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "direct_pgo.cpp"), "-L" + libdir,
                           "-lsim3opt", "-Wl,-rpath," + libdir, "-o", exe])
    return exe


def test_shim_example_compiles_and_fails_loudly_without_gpu(tmp_path):
    import torch
    exe = compile_example(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run([exe, K.FIXTURE, str(tmp_path / "o.txt"), "1", "3"], capture_output=True,
                       text=True)
    assert r.returncode == 1 and "no usable HIP device" in r.stderr


@pytest.mark.gpu
def test_shim_example_runs_direct_pgo(tmp_path):
    from sim3opt_amd import lib as L, sim3np as S3
    exe = compile_example(tmp_path)
    out = str(tmp_path / "direct_pure.txt")
    r = subprocess.run([exe, K.FIXTURE, out, "1", "3"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "chi2 169.9259622" in r.stdout
    rows = np.loadtxt(out, comments="%")
    assert rows.shape == (771, 9)
    # same run through the C-ABI directly: bit-identical poses (deterministic kernels)
    G = L.Graph()
    G.load_kitti_direct(K.FIXTURE, True)
    G.initialize()
    G.optimize(3)
    Swc = S3.inv(G.get_vertices())
    assert np.abs(rows[:, 2:5] - Swc[:, 4:7]).max() < 1e-12
    assert np.abs(rows[:, 1] - G.get_vertices()[:, 7]).max() < 1e-15
