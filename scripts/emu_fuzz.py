"""The scale fuzz of the fp16-piece training plans (tests/parity_cases.py::case_f16x3_scale_fuzz: inputs x 1e-6 ... 1e3, weights x 0.03 ... 30,
biases x 0 ... 100, cotangents 1e-33 ... 1e3, the out-of-range corner) at geometries and batch sizes the CPU suite has no time for, on the
CPU wave emulator:   python scripts/emu_fuzz.py northstar8x256 300      (one geometry per process; prints PASS / FAIL + what failed)
Test infrastructure."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import backends as B  # noqa: E402
import parity_cases as P  # noqa: E402

name, m = sys.argv[1], int(sys.argv[2])
b = B.EmuBackend()
t = time.time()
try:
    P.case_f16x3_scale_fuzz(b, m=m, names=(name,))
    print("PASS", name, "m = %d" % m, "%.0f s" % (time.time() - t), flush=True)
except AssertionError as e:
    print("FAIL", name, "m = %d" % m, "%.0f s" % (time.time() - t), str(e)[:3000], flush=True)
    sys.exit(1)
