"""The arithmetic of the shipped kernel sources at LATE-TRAINING weights, sampler taken out (companion of scripts/emu_soak.py,
DESIGN.md 8.12): the reference's own 200 000-iteration lego-lowres nets (tests/golden/lego_lowres_weights.npz), the embedded sample
points a training step really feeds them (48x48 views of the scene, 64 + 64 samples, perturbed, noise 0.2, white background) and the
cotangents d(loss)/d(raw) that step really produces (most of them tiny or exactly zero), through the MLP forward + backward of
  * the fp32 kernels, * the fp16-piece kernels (NERFHIP_PRECISION_F16X3_TRAIN), both on the CPU wave emulator (tests/emu),
  * torch fp32 (the oracle's autograd),
each against the SAME computation in float64.  Teacher-forced (inputs and cotangents fixed), on the samples that pass the ReLU-margin
filter of tests/tolerances.py -- the comparison of tests/parity_cases.py::case_mlp_backward, at weights and data it never sees.
    python scripts/emu_pretrained_grad.py > profiles/r05_emu_pretrained_grad.txt
Test infrastructure + oracle: not a product path."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "scripts")):
    sys.path.insert(0, p)

import backends as B  # noqa: E402
import emu_soak as S  # noqa: E402  (teacher views)
import nerf_oracle as O  # noqa: E402
import nerf_pytorch_amd._lib as L  # noqa: E402
import tolerances as T  # noqa: E402

CFG = B.model_cfg()
NC, NF, RAYS = 64, 64, 40


def embed(pts, rays):
    flat = pts.reshape(-1, 3)
    dirs = rays[..., None, -3:].expand(pts.shape).reshape(-1, 3)
    return torch.cat((O.positional_encoding(flat, 10, True, True), O.positional_encoding(dirs, 4, True, True)), dim=-1)


def step_data(par_c, par_f, rays, target, seed):
    """One training step of the oracle at the pretrained weights: per net, the embedded inputs and d(loss)/d(raw)."""
    g = torch.Generator().manual_seed(seed)
    n = rays.shape[0]
    rand = dict(t_rand=torch.rand(n, NC, generator=g), noise_coarse=torch.randn(n, NC, generator=g), u=torch.rand(n, NF, generator=g),
                noise_fine=torch.randn(n, NC + NF, generator=g))
    opt = dict(num_coarse=NC, num_fine=NF, perturb=True, lindisp=False, white_background=True, noise_std=0.2)
    with torch.no_grad():
        out = O.render_rays(rays, par_c, par_f, CFG, CFG, opt, rand)
    ro, rd = rays[..., :3], rays[..., 3:6]
    data = {}
    for net, par, z, noise in (("coarse", par_c, out["z_coarse"], rand["noise_coarse"]), ("fine", par_f, out["z_fine"], rand["noise_fine"])):
        x = embed(ro[..., None, :] + rd[..., None, :] * z[..., :, None], rays)
        raw = O.mlp_forward(par, x, CFG).reshape(n, -1, 4).detach().requires_grad_(True)
        rgb = O.volume_render(raw, z, rd, 0.2, noise, True)[0]
        ((rgb - target) ** 2).mean().backward()  # img2mse of this net's map (train_nerf.py:244-256)
        data[net] = (x.contiguous(), raw.grad.reshape(-1, 4).contiguous())
    return data


def grads(params, x, go, dtype):
    p = {k: v.to(dtype).clone().requires_grad_(True) for k, v in params.items()}
    y = O.mlp_forward(p, x.to(dtype), CFG)
    (y * go.to(dtype)).sum().backward()
    return y.detach(), {k: v.grad for k, v in p.items()}


def worst(got, ref):
    w, where = 0.0, ""
    for k, r in ref.items():
        r = r.numpy() if hasattr(r, "numpy") else r
        e = float(np.abs(np.asarray(got[k], np.float64) - r).max() / max(float(np.abs(r).max()), 1e-300))
        if e > w:
            w, where = e, k
    return w, where


def main():
    torch.set_num_threads(4)
    b = B.EmuBackend()
    w = np.load(S.WEIGHTS)
    nets = dict(coarse={k[2:]: torch.from_numpy(w[k]) for k in w.files if k.startswith("c_")},
                fine={k[2:]: torch.from_numpy(w[k]) for k in w.files if k.startswith("f_")})
    rays_all, imgs, train, _ = S.teacher_views()
    margin = T.bound("relu_margin")
    print("# python scripts/emu_pretrained_grad.py   (kernel sources on the CPU wave emulator; the reference's pretrained lego-lowres nets, %d rays x"
          % RAYS)
    print("# (%d + %d) samples per case, inputs and cotangents of a real training step at those weights; ReLU-margin filter %g (relative);" %
          (NC, NF, margin))
    print("# every column: worst parameter tensor's max|gradient - float64 gradient| / max|float64 gradient|; forward: max|raw - float64 raw| / max|raw|)")
    print("# largest |weight| in the two nets: %.2f / %.2f; cotangent rows exactly zero and the span of the others are listed per case"
          % tuple(max(float(v.abs().max()) for k, v in nets[n].items() if k.endswith("weight")) for n in ("coarse", "fine")))
    print("%-22s %6s %6s %22s | %-30s | %-30s | %-30s" % ("case", "kept", "zero", "nonzero |go| span", "torch fp32: fwd, grad (tensor)",
                                                         "fp32 kernels: fwd, grad", "fp16 pieces: fwd, grad"))
    summary = {"torch": [], "fp32": [], "f16": []}
    for view in train[:6]:
        gsel = torch.Generator().manual_seed(500 + view)
        pix = torch.randperm(S.SIDE * S.SIDE, generator=gsel)[:RAYS]
        data = step_data(nets["coarse"], nets["fine"], rays_all[view][pix], imgs[view][pix], seed=900 + view)
        for net in ("coarse", "fine"):
            x, go = data[net]
            keep = O.mlp_relu_margin(nets[net], x, CFG) > margin
            x, go = x[keep].contiguous(), go[keep].contiguous()
            rowmax = go.abs().amax(dim=1)
            nz = rowmax[rowmax > 0]
            y64, g64 = grads(nets[net], x, go, torch.float64)
            y32, g32 = grads(nets[net], x, go, torch.float32)
            ymax = float(y64.abs().max())
            cols = [(float((y32.double() - y64).abs().max()) / ymax,) + worst({k: v.numpy() for k, v in g32.items()}, g64)]
            for prec in (L.PRECISION_FP32, L.PRECISION_F16X3_TRAIN):
                plan = b.make_plan(CFG, prec)
                packed = b.pack(plan, b.flatten_params(plan, {k: v.numpy() for k, v in nets[net].items()}))
                y, stash = b.mlp_fwd(plan, packed, x.numpy(), want_stash=True)
                gk = b.unflatten(plan, b.mlp_bwd(plan, packed, go.numpy(), stash))
                assert np.isfinite(y).all() and all(np.isfinite(v).all() for v in gk.values())
                cols.append((float(np.abs(y - y64.numpy()).max()) / ymax,) + worst(gk, g64))
                b.lib.plan_destroy(plan)
            for nm, c in zip(("torch", "fp32", "f16"), cols):
                summary[nm].append(c[1])
            print("%-22s %6d %6d %10.1e .. %8.1e | %8.1e %8.1e %-12s | %8.1e %8.1e %-12s | %8.1e %8.1e %-12s" % (
                "view %d, %s net" % (view, net), int(keep.sum()), int((rowmax == 0).sum()), float(nz.min()), float(nz.max()),
                cols[0][0], cols[0][1], cols[0][2][:12], cols[1][0], cols[1][1], cols[1][2][:12], cols[2][0], cols[2][1], cols[2][2][:12]), flush=True)
    print("# gradient distance to float64 over the %d cases, median / max:  torch fp32 %.1e / %.1e   fp32 kernels %.1e / %.1e   fp16 pieces %.1e / %.1e"
          % (len(summary["torch"]), np.median(summary["torch"]), max(summary["torch"]), np.median(summary["fp32"]), max(summary["fp32"]),
             np.median(summary["f16"]), max(summary["f16"])))


if __name__ == "__main__":
    main()
