"""A training run of the SHIPPED kernel sources on the CPU wave emulator (tests/emu), fp32 plans next to fp16-piece plans
(NERFHIP_PRECISION_F16X3_TRAIN): the long runs of profiles/r05_psnr_soak.txt were measured on MI355X on builds that precede the
round's last three range fixes; this one needs no GPU, so it runs on the final sources -- at a size the emulator
finishes: 4x128 nets (the reference's default geometry; its hidden x hidden weight-gradient blocks run on k_wgrad_f16x3), 32 + 32
samples per ray, 32 rays per iteration (16 s per fp16-piece step on one core).

Scene: the pretrained lego-lowres nets of the reference (tests/golden/lego_lowres_weights.npz) rendered by the oracle at 48x48 on a
white background -- most rays hit nothing, i.e. exactly zero cotangents on most samples, the regime of DESIGN.md 8.8.
Both arms: same initial weights (--init random: torch's default init; --init pretrained: the teacher's own weights, i.e. the
statistics of iteration 200 000 from the first step on), views, pixels and random draws.  Every --diag iterations, on the step's own batch:
  * is the flat gradient / loss finite (checked EVERY iteration as well),
  * the oracle's autograd gradient (torch fp32 on the CPU) at the arm's weights, and the distance of the arm's kernels' gradient
    from it -- in the fp16-piece arm also of the fp32 kernels' gradient at the SAME weights --, worst tensor, as a fraction of that
    tensor's max|g|,
  * validation PSNR of the arm's weights on held-out views (oracle render, no perturbation).
Usage (one process per arm; the data stream depends on the seed only):
    python scripts/emu_soak.py --arm f16x3_train --iters 1500 --seed 1 --out profiles/r05_emu_soak_runs/seed1_f16x3_train.json
Test infrastructure + oracle: not a product path."""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import backends as B  # noqa: E402
import nerf_oracle as O  # noqa: E402
import nerf_pytorch_amd._lib as L  # noqa: E402

CFG = B.model_cfg()  # 4 x 128, skip 4, 10 / 4 frequencies, view directions: train_nerf.py:117-134's default
SIDE, FOCAL = 48, 0.5 * 48 / math.tan(0.5 * 0.6911112070083618)
NEAR, FAR = 2.0, 6.0
WEIGHTS = os.path.join(ROOT, "tests", "golden", "lego_lowres_weights.npz")  # pretrained/lego-lowres/checkpoint199999.ckpt's two nets


def pose_spherical(theta_deg, phi_deg, radius):
    """load_blender.py:34-49 restated (translate along z, rotate about x by phi, about y by theta, flip to the blender frame)."""
    th, ph = math.radians(theta_deg), math.radians(phi_deg)
    t = torch.eye(4)
    t[2, 3] = radius
    rp = torch.tensor([[1, 0, 0, 0], [0, math.cos(ph), -math.sin(ph), 0], [0, math.sin(ph), math.cos(ph), 0], [0, 0, 0, 1.0]])
    rt = torch.tensor([[math.cos(th), 0, -math.sin(th), 0], [0, 1, 0, 0], [math.sin(th), 0, math.cos(th), 0], [0, 0, 0, 1.0]])
    flip = torch.tensor([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1.0]])
    return flip @ rt @ rp @ t


def teacher_views(n_train=12, n_val=2):
    CFG = B.model_cfg()  # (the teacher's own geometry, whatever the students')
    w = np.load(WEIGHTS)
    par_c = {k[2:]: torch.from_numpy(w[k]) for k in w.files if k.startswith("c_")}
    par_f = {k[2:]: torch.from_numpy(w[k]) for k in w.files if k.startswith("f_")}
    opt = dict(num_coarse=64, num_fine=64, perturb=False, lindisp=False, white_background=True, noise_std=0.0)
    g = torch.Generator().manual_seed(2020)
    n = n_train + n_val
    thetas = (torch.linspace(-180, 180, n + 1)[:-1] + 1.3 * torch.rand(n, generator=g)).tolist()
    phis = (-50.0 + 40.0 * torch.rand(n, generator=g)).tolist()
    rays, imgs = [], []
    with torch.no_grad():
        for th, ph in zip(thetas, phis):
            ro, rd = O.get_ray_bundle(SIDE, SIDE, FOCAL, pose_spherical(th, ph, 4.0))
            r = O.pack_rays(ro, rd, NEAR, FAR, rd)
            rays.append(r)
            imgs.append(O.render_rays(r, par_c, par_f, CFG, CFG, opt)["rgb_fine"].clamp(0, 1))
    val = [3, 10][:n_val]
    train = [i for i in range(n) if i not in val]
    return torch.stack(rays), torch.stack(imgs), train, val


class Arm:
    """One student pair (coarse + fine net) on plans of one arithmetic; the training step through the C ABI."""

    def __init__(self, b, precision, seed, nc, nf, noise, init="random"):
        self.b, self.precision = b, precision
        self.opt = dict(num_coarse=nc, num_fine=nf, perturb=True, lindisp=False, white_background=True, noise_std=noise)
        self.pc, self.pf = b.make_plan(CFG, precision), b.make_plan(CFG, precision)
        if init == "pretrained":  # late training: the reference's own 200 k-iteration lego-lowres nets, trained on
            w = np.load(WEIGHTS)
            start = [{k[2:]: w[k] for k in w.files if k.startswith(pre)} for pre in ("c_", "f_")]
        else:
            start = [{k: v.numpy() for k, v in O.init_params(CFG, seed=s).items()} for s in (100 + seed, 200 + seed)]
        self.flat = [b.flatten_params(p, st) for p, st in zip((self.pc, self.pf), start)]
        self.m = [np.zeros_like(f) for f in self.flat]
        self.v = [np.zeros_like(f) for f in self.flat]
        self.step = 0

    def params(self, which):
        return {k: torch.from_numpy(np.array(v)) for k, v in self.b.unflatten((self.pc, self.pf)[which], self.flat[which]).items()}

    def gradient(self, rays, target, rand, flat=None):
        """Forward (training) -> loss -> backward on the plans of this arm at `flat` (default: the arm's own weights)."""
        b = self.b
        flat = self.flat if flat is None else flat
        packed = [b.pack(p, f) for p, f in zip((self.pc, self.pf), flat)]
        loss = []

        def cotangents(out):  # train_nerf.py:244-259: img2mse(coarse) + img2mse(fine), backward
            l3, gc, gf = b.mse_loss(out["rgb_coarse"], out["rgb_fine"], target)
            loss.append(float(l3[2]))
            return gc, gf
        out = b.render(self.pc, self.pf, packed[0], packed[1], rays, self.opt, rand, training=True, g_rgb=cotangents)
        return loss[0], out["g_params_coarse"], out["g_params_fine"]

    def train_step(self, rays, target, rand, lr):
        loss, gc, gf = self.gradient(rays, target, rand)
        finite = bool(np.isfinite(gc).all() and np.isfinite(gf).all() and math.isfinite(loss))
        self.step += 1
        for i, g in enumerate((gc, gf)):
            self.flat[i], self.m[i], self.v[i] = self.b.adam_step(self.flat[i], g, self.m[i], self.v[i], lr, self.step)
        return loss, finite, gc, gf


def oracle_gradient(par_c, par_f, rays, target, opt, rand):
    pc = {k: v.clone().requires_grad_(True) for k, v in par_c.items()}
    pf = {k: v.clone().requires_grad_(True) for k, v in par_f.items()}
    out = O.render_rays(torch.from_numpy(rays), pc, pf, CFG, CFG, opt, {k: torch.from_numpy(v) for k, v in rand.items()})
    loss, _, _, _ = O.loss_and_psnr(out["rgb_coarse"], out["rgb_fine"], torch.from_numpy(target))
    loss.backward()
    return float(loss), {k: v.grad.numpy() for k, v in pc.items()}, {k: v.grad.numpy() for k, v in pf.items()}


def worst_rel(b, plan, flat_grad, ref):
    worst, where = 0.0, ""
    for k, v in b.unflatten(plan, flat_grad).items():
        e = float(np.abs(v - ref[k]).max() / max(float(np.abs(ref[k]).max()), 1e-30))
        if e > worst:
            worst, where = e, k
    return worst, where


def val_psnr(arm, rays, imgs, val):
    opt = dict(arm.opt, perturb=False, noise_std=0.0)
    mse = 0.0
    with torch.no_grad():
        pc, pf = arm.params(0), arm.params(1)
        for i in val:
            out = O.render_rays(rays[i], pc, pf, CFG, CFG, opt)
            mse += float(((out["rgb_coarse"] - imgs[i]) ** 2).mean() + ((out["rgb_fine"] - imgs[i]) ** 2).mean())
    return -10.0 * math.log10(mse / len(val))  # train_nerf.py:339-347: PSNR of the summed coarse + fine mse


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=3000)
    ap.add_argument("--rays", type=int, default=32)
    ap.add_argument("--nc", type=int, default=32)
    ap.add_argument("--nf", type=int, default=32)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--arm", choices=("fp32", "f16x3_train"), required=True)
    ap.add_argument("--diag", type=int, default=250)
    ap.add_argument("--lr", type=float, default=5e-3)  # config/lego.yml: 5e-3, decayed by 0.1 per 250 k iterations
    ap.add_argument("--noise", type=float, default=0.2)  # config/lego.yml: radiance_field_noise_std 0.2
    ap.add_argument("--init", choices=("random", "pretrained"), default="random",
                    help="pretrained: start from the teacher's own weights -- the late-training regime (saturated densities, "
                         "tiny cotangents) from the first iteration on")
    ap.add_argument("--layers", type=int, default=4, help="students' num_layers (skip 4; --init random only): 8 makes the skip "
                    "layer's xyz columns guests of a hidden x hidden block in k_wgrad_f16x3")
    ap.add_argument("--hidden", type=int, default=128, help="students' hidden_size (--init random only): 64 = the 64-wide kernel instances")
    ap.add_argument("--skip", type=int, default=4)
    ap.add_argument("--fx", type=int, default=10, help="students' num_encoding_fn_xyz (config/fern.yml: 6)")
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    if (args.layers, args.hidden, args.skip, args.fx) != (4, 128, 4, 10):
        assert args.init == "random", "the pretrained weights are 4 x 128, skip 4, 10 frequencies"
        CFG.update(B.model_cfg(num_layers=args.layers, hidden_size=args.hidden, skip_connect_every=args.skip, num_encoding_fn_xyz=args.fx))
    torch.set_num_threads(1)
    b = B.EmuBackend()
    rays_all, imgs, train, val = teacher_views()
    prec = dict(fp32=L.PRECISION_FP32, f16x3_train=L.PRECISION_F16X3_TRAIN)[args.arm]
    arm = Arm(b, prec, args.seed, args.nc, args.nf, args.noise, args.init)
    ref32 = Arm(b, L.PRECISION_FP32, args.seed, args.nc, args.nf, args.noise) if prec else None  # (its plans only: diagnostics)
    g = torch.Generator().manual_seed(1000 + args.seed)  # the data stream: identical in every arm of a seed
    rec = dict(args=vars(args), cfg="%dx%d skip %d, %d xyz frequencies, %d + %d samples, %d rays / iteration, %dx%d views of the lego-lowres teacher, white "
               "background" % (args.layers, args.hidden, args.skip, args.fx, args.nc, args.nf, args.rays, SIDE, SIDE), lib_sources_sha16=None, checkpoints=[], nonfinite=[], losses=[])
    try:
        import bench
        rec["lib_sources_sha16"] = bench.lib_sources_sha16()
    except Exception:  # noqa: BLE001
        pass
    t0 = time.time()
    for it in range(1, args.iters + 1):
        view = train[int(torch.randint(len(train), (1,), generator=g))]
        pix = torch.randperm(SIDE * SIDE, generator=g)[:args.rays]
        rays = rays_all[view][pix].numpy().copy()
        target = imgs[view][pix].numpy().copy()
        n = args.rays
        rand = dict(t_rand=torch.rand(n, args.nc, generator=g).numpy(), noise_coarse=torch.randn(n, args.nc, generator=g).numpy(),
                    u=torch.rand(n, args.nf, generator=g).numpy(), noise_fine=torch.randn(n, args.nc + args.nf, generator=g).numpy())
        lr = args.lr * 0.1 ** (it / 250000.0)
        diag = it % args.diag == 0 or it == 1
        if diag:  # the oracle's and the fp32 kernels' gradient at this arm's weights, BEFORE its step
            _, oc, of = oracle_gradient(arm.params(0), arm.params(1), rays, target, arm.opt, rand)
            if ref32 is not None:
                _, kc32, kf32 = ref32.gradient(rays, target, rand, flat=arm.flat)
        loss, finite, gc, gf = arm.train_step(rays, target, rand, lr)
        rec["losses"].append(round(loss, 6))
        if not finite:
            rec["nonfinite"].append(it)
        if diag:
            cp = dict(iteration=it, loss=loss, seconds=round(time.time() - t0, 1), val_psnr=round(val_psnr(arm, rays_all, imgs, val), 3),
                      grad_vs_oracle={args.arm: dict(coarse=worst_rel(b, arm.pc, gc, oc), fine=worst_rel(b, arm.pf, gf, of))},
                      grad_max=dict(coarse=float(np.abs(gc).max()), fine=float(np.abs(gf).max())),
                      background_rays=int((target.min(axis=1) >= 0.999).sum()))
            if ref32 is not None:
                cp["grad_vs_oracle"]["fp32_kernels_same_weights"] = dict(coarse=worst_rel(b, arm.pc, kc32, oc),
                                                                         fine=worst_rel(b, arm.pf, kf32, of))
            rec["checkpoints"].append(cp)
            print(json.dumps(cp), flush=True)
        if diag or it % 25 == 0:
            rec["seconds"] = round(time.time() - t0, 1)
            os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
            json.dump(rec, open(args.out + ".tmp", "w"))
            os.replace(args.out + ".tmp", args.out)


if __name__ == "__main__":
    main()
