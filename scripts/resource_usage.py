"""Register / scratch / LDS / occupancy table of every gfx950 kernel (clang -Rpass-analysis=kernel-resource-usage):
    python scripts/resource_usage.py > profiles/rNN_resource_usage.txt"""
import os
import re
import subprocess

HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nerf-pytorch_amd", "csrc")
FLAGS = ("--offload-arch=gfx950 -x hip -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -pragma-unroll-threshold=4000000 "
         "-Rpass-analysis=kernel-resource-usage").split()
KEYS = ("TotalSGPRs", "VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill", "VGPRs Spill",
        "LDS Size [bytes/block]")
print("%-78s %5s %5s %5s %7s %4s %6s %6s %6s" % ("kernel", "SGPR", "VGPR", "AGPR", "scratch", "occ", "sspill", "vspill",
                                                 "LDS"))
import sys
FILES = sys.argv[1:] or (
    "mlp16.hip", "mlp16_w512.hip", "mlp16_ext.hip", "mlp64r.hip", "mlp_f16w.hip", "wgrad_f16.hip", "pack_f16.hip", "wgrad.hip", "mlp.hip", "render.hip", "sample.hip", "compact.hip", "elementwise.hip", "dataio.hip")
for f in FILES:
    extra = ["-fno-slp-vectorize", "-mllvm", "-instcombine-max-copied-from-constant-users=4000"] if f == "wgrad.hip" else []  # (as the Makefile does)
    err = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-c", f, "-o", "/dev/null"], cwd=HERE, capture_output=True,
                         text=True).stderr
    cur = None
    for line in err.splitlines():
        m = re.search(r"remark:\s+(.*?):\s+(\S+) \[-Rpass", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            name = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
            name = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            cur = {"name": name}
        elif cur is not None and k in KEYS:
            cur[k] = v
            if k == KEYS[-1]:
                print("%-78s %5s %5s %5s %7s %4s %6s %6s %6s" % ((cur["name"][:78],) + tuple(cur[x] for x in KEYS)))
                cur = None
