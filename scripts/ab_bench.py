"""A/B tuning aid: run bench.py's main() on a variant build of the HIP library (make -C nerf-pytorch_amd/csrc variant NAME=... DEFS=...).
    python scripts/ab_bench.py NAME [bench.py arguments]      (NAME = "" -> the product library)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nerf_pytorch_amd._lib as L  # noqa: E402

name = sys.argv[1]
if name:
    L.LIB_PATH = os.path.join(ROOT, "nerf-pytorch_amd", "libnerfhip_%s.so" % name)
    L.ALLOW_DIAG = True  # (variant builds carry the NH_DIAG mark: the package itself refuses them)
sys.argv = ["bench.py"] + sys.argv[2:]
import bench  # noqa: E402

bench.main()
