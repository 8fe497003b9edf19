"""Diagnostic (A/B build with -DNH_PHASE_TIMING copied over libnerfhip.so by scripts/fern_ab.sh): shader cycles per phase of the fused
64-wide backward (csrc/mlp64r.hip), summed over the waves of each role, for a few fern training steps."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402,F401
import nerf_pytorch_amd as N  # noqa: E402
from nerf_pytorch_amd import _lib as L  # noqa: E402

lib = ctypes.CDLL(L.LIB_PATH)
wl = bench.WORKLOADS["fern"] if hasattr(bench, "WORKLOADS") else None
cfg = dict(num_layers=4, hidden_size=64, skip_connect_every=3, num_encoding_fn_xyz=6, num_encoding_fn_dir=4)
mc, mf = N.FlexibleNeRFModel(**cfg).cuda(), N.FlexibleNeRFModel(**cfg).cuda()
eng = N.TrainEngine(mc, mf, 64, 64, perturb=True, white_background=False, noise_std=1.0, lr=5e-3, seed=1, backward=(sys.argv[1] if len(sys.argv) > 1 else "fused"), overlap=False)
n = 4096
g = torch.Generator(device="cuda").manual_seed(1)
ro = torch.rand(n, 3, generator=g, device="cuda") - 0.5
rd = torch.rand(n, 3, generator=g, device="cuda") - 0.5
rd[:, 2] = -1.0
vd = rd / rd.norm(dim=-1, keepdim=True)
rays = torch.cat([ro, rd, torch.zeros(n, 1, device="cuda"), torch.ones(n, 1, device="cuda"), vd], dim=-1).contiguous()
tgt = torch.rand(n, 3, generator=g, device="cuda")
for i in range(3):
    eng.step_rays(rays, tgt) if hasattr(eng, "step_rays") else eng.step(rays, tgt)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
lib.nerfhip_debug_phases64(buf, 1)
steps = 5
for i in range(steps):
    eng.step_rays(rays, tgt) if hasattr(eng, "step_rays") else eng.step(rays, tgt)
torch.cuda.synchronize()
lib.nerfhip_debug_phases64(buf, 0)
v = [int(x) for x in buf]
names = ["chain: hand-over read", "chain: forward", "chain: transposed layers", "chain: wait free", "chain: put+publish", "", "", "",
         "wgrad: wait step a", "wgrad: multiply", "wgrad: wait steps b..", "wgrad: prepare next round", "", "", "", ""]
rounds = steps * (n * 64 + n * 128) / 64.0   # rounds over all workgroups
for nm, x in zip(names, v):
    if nm:
        print("%-28s %8.0f cycles per round and wave" % (nm, x / (rounds * 4)))
print("chain total %.0f, wgrad total %.0f cycles per round" % (sum(v[:8]) / (rounds * 4), sum(v[8:]) / (rounds * 4)))
