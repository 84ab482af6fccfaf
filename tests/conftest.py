import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# this harness uses torch next to the library in one process (gpu_available, torch.distributed): torch's
# HIP runtime must be the one that is mapped first (sim3opt_amd.lib.load)
os.environ.setdefault("SIM3OPT_PRELOAD_TORCH", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Both native libraries are built in-tree before any test runs (hipcc cross-compiles
    gfx950 without a GPU; the oracle needs only gcc)."""
    from oracle import oracle as O
    from sim3opt_amd import build as B

    O.build()
    B.build()
    yield


def gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
