"""GPU-box diagnostic: where a wave's cycles go inside k_mlp_fwd16 / k_mlp_dgrad16 (instrumented build
libnerfhip_dbg.so = `make -C nerf-pytorch_amd/csrc dbg`, compiled with -DNH_PHASE_TIMING: shader-clock stamps around
each phase of gemm16, summed over all waves)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import nerf_pytorch_amd as N  # noqa: E402

N._lib.LIB_PATH = os.path.join(ROOT, "nerf-pytorch_amd", "libnerfhip_dbg.so")

dev = torch.device("cuda", 0)
lib = N._lib.get_lib()
cfg = dict(num_layers=8, hidden_size=256, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)
torch.manual_seed(0)
mc, mf = N.FlexibleNeRFModel(**cfg).to(dev), N.FlexibleNeRFModel(**cfg).to(dev)
eng = N.TrainEngine(mc, mf, 64, 128, noise_std=0.2)
n = 4096
g = torch.Generator().manual_seed(0)
ro = torch.tensor([0.0, 0.0, 4.0]).expand(n, 3).contiguous().to(dev)
rd = (torch.randn(n, 3, generator=g) * 0.3)
rd[:, 2] = -1.0
rays = N.pack_rays(ro, rd.to(dev), N.make_options())
tgt = torch.rand(n, 3, generator=g).to(dev)
for _ in range(2):
    eng.step(rays, tgt)
torch.cuda.synchronize()
dbg16 = lib._dll.nerfhip_debug_phases16
dbg16.argtypes = [C.c_void_p, C.c_int]
b16 = (C.c_ulonglong * 16)()
dbg16(b16, 1)
for _ in range(3):
    eng.step(rays, tgt)
torch.cuda.synchronize()
dbg16(b16, 0)
n16 = ["0 copy set-up + bias", "1 operand reads + MFMAs (+ copy pieces, stores)", "2 s_waitcnt vmcnt(0)", "3 s_barrier",
       "4 between gemms (epilogue, encodings)"]
for base, k in ((0, "k_mlp_fwd16"), (8, "k_mlp_dgrad16")):
    v = [b16[base + i] for i in range(5)]
    tot = float(sum(v))
    print(k, "total wave-cycles %.3e (3 steps)" % tot)
    for i in range(5):
        print("   %-50s %6.2f %%" % (n16[i], 100.0 * v[i] / tot))
