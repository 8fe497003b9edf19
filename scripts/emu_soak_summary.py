"""Summary of scripts/emu_soak.py runs (one JSON per arm):   python scripts/emu_soak_summary.py profiles/r05_emu_soak_runs > profiles/r05_emu_soak.txt"""
import glob
import json
import os
import sys

import numpy as np

d = sys.argv[1]
runs = {}
for path in sorted(glob.glob(os.path.join(d, "*.json"))):
    name = os.path.basename(path)[:-5]
    for arm in ("f16x3_train", "fp32"):
        if name.endswith("_" + arm):
            runs.setdefault(name[:-len(arm) - 1], {})[arm] = json.load(open(path))

print("# scripts/emu_soak.py: the SHIPPED kernel sources (tests/emu: every kernel of the step executed by the CPU wave emulator) trained for")
print("# real: 4x128 nets (one pair 8x128; three pairs 4x64 skip 3 with 6 frequencies = config/fern.yml's declared geometry, on the 64-wide")
print("# kernel instances), 32 + 32 samples per ray, 32 rays per iteration, Adam, the lego-lowres teacher at 48x48 on a white background;")
print("# arms fp32 plans / NERFHIP_PRECISION_F16X3_TRAIN plans: same initial weights, views, pixels, draws.  `random`: torch's default init,")
print("# lr 5e-3 (early training); `pretrained`: started from the reference's own 200 000-iteration lego-lowres weights at that iteration's")
print("# lr 7.9e-4 (late training: saturated densities, background rays with exactly zero cotangents).  Gradient columns: worst tensor's")
print("# max|kernel - oracle autograd| / max|oracle| on the step's own batch at the arm's weights (UNFILTERED: a ReLU branch or a fine sample")
print("# that round-off places differently shows as 1e-4 .. 1e-2 in either arithmetic); f16 arm: next to it the fp32 kernels' at the same weights.")
for key, arms in runs.items():
    a32, a16 = arms.get("fp32"), arms.get("f16x3_train")
    print()
    any_arm = a16 or a32
    print("## %s   (kernel sources %s; %s)" % (key, any_arm.get("lib_sources_sha16"), any_arm["cfg"]))
    for nm, a in (("fp32", a32), ("f16x3_train", a16)):
        if a:
            print("%-12s iterations done %5d   non-finite gradients / losses: %d   wall %.0f s" % (nm, len(a["losses"]), len(a["nonfinite"]),
                                                                                                   a.get("seconds", 0)))
    if a32 and a16:
        n = min(len(a32["losses"]), len(a16["losses"]))
        l32, l16 = np.array(a32["losses"][:n]), np.array(a16["losses"][:n])
        print("loss, mean over windows of 100 iterations (fp32 | f16x3_train | ratio):")
        for lo in range(0, n - n % 100, 100):
            m32, m16 = l32[lo:lo + 100].mean(), l16[lo:lo + 100].mean()
            print("   %4d-%4d   %.5f | %.5f | %.3f" % (lo + 1, lo + 100, m32, m16, m16 / m32))
        same = int((np.abs(l32 - l16) <= 1e-6 * np.maximum(np.abs(l32), 1e-12)).sum())
        print("   first iteration whose loss differs by more than 1e-6 relative: %s" %
              (int(np.argmax(np.abs(l32 - l16) > 1e-6 * np.abs(l32))) + 1 if same < n else "none"))
    print("%6s | %-19s | %-19s | %-25s | %-51s" % ("iter", "val PSNR fp32 / f16", "batch loss fp32 / f16", "fp32 arm: grad vs oracle",
                                                    "f16 arm: grad vs oracle (f16 kernels ; fp32 kernels)"))
    print("%6s | %-19s | %-19s | %-25s | %-51s" % ("", "", "", "coarse / fine", "coarse / fine ; coarse / fine"))
    c32 = {c["iteration"]: c for c in (a32["checkpoints"] if a32 else [])}
    c16 = {c["iteration"]: c for c in (a16["checkpoints"] if a16 else [])}
    worst = dict(f16=0.0, k32_same=0.0, fp32arm=0.0)
    for it in sorted(set(c32) | set(c16)):
        x, y = c32.get(it), c16.get(it)
        f = lambda c, arm, net: c["grad_vs_oracle"][arm][net][0] if c and arm in c["grad_vs_oracle"] else float("nan")
        g32 = (f(x, "fp32", "coarse"), f(x, "fp32", "fine"))
        g16 = (f(y, "f16x3_train", "coarse"), f(y, "f16x3_train", "fine"))
        k32 = (f(y, "fp32_kernels_same_weights", "coarse"), f(y, "fp32_kernels_same_weights", "fine"))
        print("%6d | %8s / %8s | %8s / %8s | %9.1e / %9.1e | %9.1e / %9.1e ; %9.1e / %9.1e" % (
            it, "%.3f" % x["val_psnr"] if x else "-", "%.3f" % y["val_psnr"] if y else "-",
            "%.5f" % x["loss"] if x else "-", "%.5f" % y["loss"] if y else "-", g32[0], g32[1], g16[0], g16[1], k32[0], k32[1]))
    if a16:
        g = np.array([[c["grad_vs_oracle"]["f16x3_train"][n][0] for n in ("coarse", "fine")] +
                      [c["grad_vs_oracle"]["fp32_kernels_same_weights"][n][0] for n in ("coarse", "fine")] for c in a16["checkpoints"]])
        print("f16 arm, %d diagnostics: median distance to the oracle, f16 kernels coarse / fine %.1e / %.1e; fp32 kernels at the same weights "
              "%.1e / %.1e" % ((len(g),) + tuple(np.median(g, axis=0))))
        print("   diagnostics where the f16 kernels are farther from the oracle than the fp32 kernels by more than 2x AND more than 1e-4: "
              "coarse %d, fine %d; the other way round: coarse %d, fine %d" % (
                  int(((g[:, 0] > 2 * g[:, 2]) & (g[:, 0] > 1e-4)).sum()), int(((g[:, 1] > 2 * g[:, 3]) & (g[:, 1] > 1e-4)).sum()),
                  int(((g[:, 2] > 2 * g[:, 0]) & (g[:, 2] > 1e-4)).sum()), int(((g[:, 3] > 2 * g[:, 1]) & (g[:, 3] > 1e-4)).sum())))
