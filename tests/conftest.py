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


def pytest_sessionfinish(session, exitstatus):
    """Measured quantities the parity cases noted (tests/parity_cases.py RECORD) -> gpurun_out/ (copied to profiles/)."""
    import json
    try:
        import parity_cases as P
    except Exception:
        return
    if not P.RECORD or not any(k.endswith("_gpu") for k in P.RECORD):
        return
    try:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_small_cases.json"), "w") as f:
            json.dump(P.RECORD, f, indent=1, sort_keys=True)
    except OSError:
        pass


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
