import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def gold(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


@pytest.fixture(scope="session")
def emu():
    import backends
    return backends.EmuBackend()


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    import backends
    import nerf_pytorch_amd._lib as L
    if not os.path.exists(L.LIB_PATH):  # test convenience only: the product never builds or falls back by itself
        import subprocess
        subprocess.run(["make", "-C", backends.CSRC, "lib", "-j8"], check=True)
    return backends.GpuBackend()
