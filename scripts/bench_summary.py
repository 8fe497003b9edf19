"""Prints the per-kernel table of bench.py JSON lines (stdin), one short block per line; argv[1:] = labels."""
import json
import sys

for i, line in enumerate(l for l in sys.stdin if l.startswith("{")):
    d = json.loads(line)
    r = d["roofline"]
    label = sys.argv[1 + i] if len(sys.argv) > 1 + i else ""
    print("%-40s rays/s %9.0f  ms %7.3f  step frac %.4f | %s %.4f" % (label, d["value"], d["ms_per_step"],
          d["step_frac_of_fp32_mfma_peak"], r["kernel"], r["frac"]))
    print("     ", {k: v for k, v in r["kernel_ms_per_step"].items() if v > 0.1})
