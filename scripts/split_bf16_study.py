"""Precision feasibility of a split-bf16 MFMA path, BEFORE any kernel is written.

The only way past the fp32-MFMA ceiling (157 TF; PyTorch-ROCm already runs the workload at ~29 % of it, so exact fp32 caps
the speed-up at ~3.4x) is v_mfma_f32_32x32x16_bf16 (2.5 PF dense) on operands split into bf16 pieces with fp32
accumulation:  a = a0 + a1 (+ a2),  a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1).
    bf16x3:  a.b ~ a0 b0 + a0 b1 + a1 b0                              (own roofline 2.5 PF / 3 = 833 TF)
    bf16x6:  ... + a1 b1 + a0 b2 + a2 b0                              (2.5 PF / 6 = 417 TF = 2.65x the fp32 pipe)
Products of two bf16 values are exact in fp32, so running the reference's torch ops with every nn.Linear replaced by
the sum of fp32 matmuls of the split pieces reproduces that arithmetic (up to accumulation order) without a kernel.
This script renders a full batch that way and measures the outputs against the fp32 oracle with the statistics of
tests/test_gpu_fullsize.py -- the same table the fp32 HIP path and the reference's own GPU path are judged by -- i.e.
whether the north star's 1e-4 bar survives the split at all.

    python scripts/split_bf16_study.py [rays] [device]        (runs on the CPU; a GPU only makes it faster)
"""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import nerf_oracle as O  # noqa: E402

TERMS = {"bf16x1": ((0, 0),), "bf16x3": ((0, 0), (0, 1), (1, 0)), "bf16x6": ((0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0))}


def split3(t):
    a0 = t.bfloat16().float()
    a1 = (t - a0).bfloat16().float()
    a2 = (t - a0 - a1).bfloat16().float()
    return (a0, a1, a2)


def make_linear(mode):
    terms = TERMS[mode]

    def linear(x, w, b=None):
        xs, ws = split3(x), split3(w)
        acc = None
        for i, j in terms:
            y = torch.matmul(xs[i], ws[j].t())
            acc = y if acc is None else acc + y
        return acc if b is None else acc + b
    return linear


def stats(got, want, bar=1e-4):
    e = np.abs(np.nan_to_num(got.double().numpy()) - np.nan_to_num(want.double().numpy()))
    per_ray = e.reshape(e.shape[0], -1).max(axis=1)
    return dict(max=float(e.max()), p999=float(np.quantile(e, 0.999)), mean=float(e.mean()), rays_over_1e4=int((per_ray > bar).sum()))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    dev = torch.device(sys.argv[2]) if len(sys.argv) > 2 else torch.device("cpu")
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    import parity_cases as P
    out = {}
    cases = [("lego batch, 8x256 random init, 64+128, noise 0.2 (tests/test_gpu_fullsize.py lego)", P.MLP_GEOMETRIES["northstar8x256"],
              None, 64, 128, 0.2, False, True)]
    w = np.load(os.path.join(ROOT, "tests", "golden", "lego_lowres_weights.npz"))
    trained = ({k[2:]: torch.from_numpy(w[k]) for k in w.files if k.startswith("c_")},
               {k[2:]: torch.from_numpy(w[k]) for k in w.files if k.startswith("f_")})
    cases.append(("pretrained lego-lowres nets (4x128), 64+64, deterministic, white background", P.MLP_GEOMETRIES["default4x128"], trained,
                  64, 64, 0.0, True, False))
    real_linear = F.linear
    for name, cfg, params, nc, nf, noise, white, perturb in cases:
        g = torch.Generator().manual_seed(101)
        pc, pf = params if params is not None else (O.init_params(cfg, seed=102), O.init_params(cfg, seed=103))
        ro = torch.tensor([0.0, 0.0, 4.0]).expand(n, 3) + 0.02 * torch.randn(n, 3, generator=g)
        rd = torch.randn(n, 3, generator=g) * 0.35
        rd[:, 2] = -1.0
        rays = O.pack_rays(ro, rd, 2.0, 6.0, rd).to(dev)
        rand = dict(t_rand=torch.rand(n, nc, generator=g), noise_coarse=torch.randn(n, nc, generator=g),
                    u=torch.rand(n, nf, generator=g), noise_fine=torch.randn(n, nc + nf, generator=g))
        rand = {k: v.to(dev) for k, v in rand.items()} if perturb else None
        opt = dict(num_coarse=nc, num_fine=nf, perturb=perturb, lindisp=False, white_background=white, noise_std=noise)
        pc = {k: v.to(dev) for k, v in pc.items()}
        pf = {k: v.to(dev) for k, v in pf.items()}
        res = {}
        with torch.no_grad():
            want = O.render_rays(rays, pc, pf, cfg, cfg, opt, rand, chunksize=65536)
            want64 = O.render_rays(rays.double(), {k: v.double() for k, v in pc.items()}, {k: v.double() for k, v in pf.items()},
                                   cfg, cfg, opt, None if rand is None else {k: v.double() for k, v in rand.items()}, chunksize=65536)
            res["fp32 vs fp64 (the reference against itself)"] = {k: stats(want[k].cpu(), want64[k].cpu()) for k in
                                                                   ("rgb_coarse", "rgb_fine", "acc_fine", "depth_fine")}
            for mode in ("bf16x6", "bf16x3", "bf16x1"):
                t0 = time.perf_counter()
                O.F.linear = make_linear(mode)
                try:
                    got = O.render_rays(rays, pc, pf, cfg, cfg, opt, rand, chunksize=65536)
                finally:
                    O.F.linear = real_linear
                res[mode + " vs fp32"] = {k: stats(got[k].cpu(), want[k].cpu()) for k in ("rgb_coarse", "rgb_fine", "acc_fine", "depth_fine")}
                res[mode + " vs fp32"]["raw_coarse_max_abs"] = float((got["raw_coarse"] - want["raw_coarse"]).abs().max())
                res[mode + " vs fp32"]["seconds"] = round(time.perf_counter() - t0, 1)
        out[name] = dict(rays=n, results=res)
        print("==", name, "(%d rays)" % n)
        for k, v in res.items():
            print("  %-46s" % k + "  ".join("%s max %.1e p99.9 %.1e over-1e-4 %d |" % (o[:9], s["max"], s["p999"], s["rays_over_1e4"])
                                          for o, s in v.items() if isinstance(s, dict)))
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "split_bf16_study.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
