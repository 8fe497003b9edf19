"""Teacher-forced MLP backward of the fp16-piece training plans (tests/parity_cases.py::case_mlp_backward, the fp32 kernels' own bound)
at every geometry the plans exist for, every plan level, m = 400, plus one end-to-end render with gradients at the 8x256 north-star
geometry -- on the CPU wave emulator, in as many processes as there are cores (the CPU suite runs a subset at m = 100 ... 120):
    python scripts/emu_sweep.py <part> <parts>      (prints PASS / FAIL lines)
Test infrastructure."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import backends as B  # noqa: E402
import parity_cases as P  # noqa: E402

part, parts = int(sys.argv[1]), int(sys.argv[2])
b = B.EmuBackend()
jobs = [("mlp_backward", name, lvl) for name in P.F16X3_GEOMETRIES for lvl in ("F16X3_FWD", "F16X3_FWD_DGRAD", "F16X3_TRAIN")]
jobs += [("render", "northstar8x256", "F16X3_TRAIN"), ("render", "fern8x128_skip3_L6", "F16X3_TRAIN"), ("input_grad", "northstar8x256", "F16X3_TRAIN")]
for i, (kind, name, lvl) in enumerate(jobs):
    if i % parts != part:
        continue
    t = time.time()
    try:
        if kind == "mlp_backward":
            P.case_mlp_backward(b, names=(name,), m=400, precision=getattr(P, lvl))
        elif kind == "input_grad":
            P.case_mlp_input_grad(b, names=(name,), m=100, precision=getattr(P, lvl))
        else:
            P.case_render_vs_oracle(b, P.MLP_GEOMETRIES[name], n=16, nc=16, nf=32, with_grads=True, tag="sweep_" + name,
                                    precision=getattr(P, lvl))
        print("PASS %-12s %-24s %-16s %5.0f s" % (kind, name, lvl, time.time() - t), flush=True)
    except AssertionError as e:
        print("FAIL %-12s %-24s %-16s %5.0f s  %s" % (kind, name, lvl, time.time() - t, str(e)[:600].replace("\n", " ")), flush=True)
