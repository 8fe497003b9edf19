"""GPU-box diagnostic: where a tile's time goes inside k_mlp_fwd / k_mlp_dgrad (instrumented build libnerfhip_dbg.so,
compiled with -DNH_PHASE_TIMING: shader-clock stamps around each phase of gemm_layer, summed over all waves)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["NERFHIP_LIB_PATH"] = os.path.join(ROOT, "nerf-pytorch_amd", "libnerfhip_dbg.so")
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import nerf_pytorch_amd as N  # noqa: E402

dev = torch.device("cuda", 0)
lib = N._lib.get_lib()
dbg = lib._dll.nerfhip_debug_phases
dbg.argtypes = [C.c_void_p, C.c_int]
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
buf = (C.c_ulonglong * 32)()
dbg(buf, 1)
for _ in range(3):
    eng.step(rays, tgt)
torch.cuda.synchronize()
dbg(buf, 0)
if os.environ.get("NERFHIP_MLP") != "32":
    dbg16 = lib._dll.nerfhip_debug_phases16
    dbg16.argtypes = [C.c_void_p, C.c_int]
    b16 = (C.c_ulonglong * 16)()
    dbg16(b16, 0)
    n16 = ["0 copy issue + stores + bias", "1 operand reads + MFMAs", "2 s_waitcnt vmcnt(0)", "3 s_barrier", "4 between gemms (epilogue, encodings)"]
    # MFMA cycles one wave issues per launch pair (coarse+fine), to compare with phase 1
    for base, k in ((0, "k_mlp_fwd16"), (8, "k_mlp_dgrad16")):
        v = [b16[base + i] for i in range(5)]
        tot = float(sum(v))
        print(k, "total wave-cycles %.3e (5 steps incl. warm-up)" % tot)
        for i in range(5):
            print("   %-44s %6.2f %%" % (n16[i], 100.0 * v[i] / tot))
    sys.exit(0)
names = ["0 dma-issue+prev-stores+inter-layer", "1 bias/operand reads + MFMA issue", "2 epilogue (drains last MFMA)",
         "3 s_waitcnt vmcnt(0)", "4 s_barrier", "5 kernel tail"]
for base, k in ((0, "k_mlp_fwd"), (8, "k_mlp_dgrad")):
    v = [buf[base + i] for i in range(6)]
    tot = float(sum(v))
    print(k, "total wave-cycles %.3e" % tot)
    for i in range(6):
        print("   %-40s %6.2f %%" % (names[i], 100.0 * v[i] / tot))
