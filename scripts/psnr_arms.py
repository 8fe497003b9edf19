"""GPU-box experiment (round 3): is the engine arm's PSNR biased against the reference's?  (VERDICT r2 weak #5.)

Round 2 (profiles/r02_psnr_400.txt; 8x256 students, 2 usable seeds) left the engine arm below the reference arm in 11 of
12 cells.  The arms differ from the reference in: kernel arithmetic (drop-in arm: ABOVE the reference in most cells),
the in-kernel Philox / Box-Muller draws, the fused Adam + loss kernels.  This script separates them with FOUR arms on
MANY seeds, everything else identical per seed (initial weights, training views, pixel draws, lr schedule):

    ref        the reference's own PyTorch path on this GPU (oracle torch ops) + torch.optim.Adam
    dropin     this package behind the reference API + torch.optim.Adam; torch's draws (the numbers `ref` consumes)
    engine_td  TrainEngine (fused loss, k_adam) FED torch's draws -- differs from `dropin` only in Adam / loss kernels
    engine     TrainEngine with its in-kernel Philox draws -- differs from `engine_td` only in the random numbers
    engine_f16fd / engine_f16tr   (round 4) `engine` with NERFHIP_PRECISION_F16X3_FWD_DGRAD / _TRAIN nets: the forward + data-gradient
               chain / every GEMM of the step on fp16 pieces (fp32-grade products; same draws as `engine`: differs from it only in the
               arithmetic).  (Round 3's bf16-piece arms -- engine_bf16fwd / _bf16fd / _bf16tr, profiles/r03_psnr_*.txt -- went with
               their kernels in round 5.)
--lr: the initial learning rate of EVERY arm (default the reference recipe's 5e-3, config/lego.yml; at 8x256 half the seeds collapse
under it in every arm -- a dead net renders a constant --, so the round-4 study of the metric's own geometry lowers it for all arms).

Scene: the teacher of scripts/psnr400.py (pretrained lego-lowres nets rendered at 400x400, 100 training / 10 held-out
views).  Students: --hidden x --layers nets (default the reference's own 4x128: its scripts build FlexibleNeRFModel with
the default sizes), 4096 rays/iter, 64+128, perturb, noise 0.2, white background, lr 5e-3 * 0.1^(i/250000).
PSNR = -10 log10(coarse_mse + fine_mse) (train_nerf.py:258-260) on 3 whole held-out views at fixed iterations.

    python scripts/psnr_arms.py SEED ITERS OUT.json [--arms a,b,..] [--hidden 128 --layers 4]
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "scripts")):
    sys.path.insert(0, p)
import nerf_oracle as O  # noqa: E402
import nerf_pytorch_amd as N  # noqa: E402
import psnr400 as P4  # noqa: E402  (teacher dataset, validation renders)

dev = torch.device("cuda", 0)
H = W = 400
FOCAL = P4.FOCAL
NC, NF, RAYS = 64, 128, 4096


def data_stream(poses, imgs, train, seed):
    """train_nerf.py:203-227: a random training view and 4096 distinct random pixels of it -- identical in every arm
    (host generator for the view, device generator for the pixels: no host round trip per step)."""
    g = torch.Generator().manual_seed(1000 + seed)
    gd = torch.Generator(device=dev).manual_seed(2000 + seed)
    while True:
        v = train[int(torch.randint(len(train), (1,), generator=g))]
        pix = torch.randperm(H * W, generator=gd, device=dev)[:RAYS]
        ro, rd = N.get_rays_at_pixels(H, W, FOCAL, poses[v][:3, :4], pix)
        yield ro, rd, imgs[v].reshape(-1, 3)[pix].contiguous()


def torch_draws(n):
    """The reference's draws per ray chunk, in its order (train_utils.py:63, volume_rendering_utils.py:30,
    nerf_helpers.py:279, volume_rendering_utils.py:30) -- what _predict_fused draws too."""
    return (torch.rand((n, NC), dtype=torch.float32, device=dev), torch.randn((n, NC), dtype=torch.float32, device=dev),
            torch.rand((n, NF), dtype=torch.float32, device=dev), torch.randn((n, NC + NF), dtype=torch.float32, device=dev))


LR0 = 5e-3


def lr_at(i):
    return N.TrainEngine.lr_at(i, lr0=LR0)


def run(arm, seed, iters, check, student, poses, imgs, train, views):
    P4.STUDENT.clear()
    P4.STUDENT.update(student)  # (validate_ref reads it)
    torch.manual_seed(seed)
    mc, mf = N.FlexibleNeRFModel(**student), N.FlexibleNeRFModel(**student)   # nn.Linear default init, reference order
    stream = data_stream(poses, imgs, train, seed)
    hist, losses = {}, []
    t_train, t_mark = 0.0, time.perf_counter()
    opts = N.make_options(NC, NF, white_background=True)
    if arm == "ref":
        pc = {k: v.detach().clone().to(dev).requires_grad_(True) for k, v in mc.state_dict().items()}
        pf = {k: v.detach().clone().to(dev).requires_grad_(True) for k, v in mf.state_dict().items()}
        opt = torch.optim.Adam(list(pc.values()) + list(pf.values()), lr=LR0)
    elif arm == "dropin":
        mc, mf = mc.to(dev), mf.to(dev)
        opt = torch.optim.Adam(list(mc.parameters()) + list(mf.parameters()), lr=LR0)
        ex, ed = N.get_embedding_function(10, True, True), N.get_embedding_function(4, True, True)
    else:
        mc, mf = mc.to(dev), mf.to(dev)
        if arm in ("engine_f16fd", "engine_f16tr"):  # the engine arm on fp16 pieces
            prec = {"engine_f16fd": "f16x3_fwd_dgrad", "engine_f16tr": "f16x3_train"}[arm]
            mc.set_training_precision(prec)
            mf.set_training_precision(prec)
        eng = N.TrainEngine(mc, mf, NC, NF, perturb=True, white_background=True, noise_std=0.2, lr=LR0, seed=seed)
    torch.manual_seed(seed + 12345)  # the draws of the training loop: arms ref / dropin / engine_td consume the same numbers
    for i in range(1, iters + 1):
        ro, rd, tgt = next(stream)
        if arm == "ref":
            rays = O.pack_rays(ro, rd, 2.0, 6.0, rd)
            d = torch_draws(rays.shape[0])
            out = O.render_rays(rays, pc, pf, student, student, P4.OPT, dict(t_rand=d[0], noise_coarse=d[1], u=d[2], noise_fine=d[3]),
                                chunksize=131072)
            loss = torch.nn.functional.mse_loss(out["rgb_coarse"], tgt) + torch.nn.functional.mse_loss(out["rgb_fine"], tgt)
            loss.backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
            losses.append(loss.detach().reshape(1))
        elif arm == "dropin":
            out = N.run_one_iter_of_nerf(H, W, FOCAL, mc, mf, ro, rd, opts, mode="train", encode_position_fn=ex,
                                         encode_direction_fn=ed)
            loss = N.img2mse(out[0], tgt) + N.img2mse(out[3], tgt)
            loss.backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
            losses.append(loss.detach().reshape(1))
        else:
            rays = N.pack_rays(ro, rd, opts)
            draws = torch_draws(rays.shape[0]) if arm == "engine_td" else None
            loss3 = eng.step(rays, tgt, lr=lr_at(i - 1), draws=draws)
            losses.append(loss3[2:3].clone())
        if arm in ("ref", "dropin"):
            for gq in opt.param_groups:  # train_nerf.py:264-270
                gq["lr"] = lr_at(i)
        if i in check:
            torch.cuda.synchronize()
            t_train += time.perf_counter() - t_mark
            recent = torch.cat(losses[-50:]).cpu().numpy()
            vals = P4.validate_ref(pc, pf, poses, imgs, views) if arm == "ref" else P4.validate_hip(mc, mf, poses, imgs, views)
            vc, vf = float(np.mean([a for a, _ in vals])), float(np.mean([b for _, b in vals]))
            hist[i] = dict(train_psnr=P4.psnr(float(np.mean(recent))), val_psnr=P4.psnr(vc + vf), val_psnr_fine=P4.psnr(vf),
                           val_psnr_coarse=P4.psnr(vc), train_wall_s=round(t_train, 2))
            print(arm, seed, i, hist[i], flush=True)
            losses = losses[-50:]
            t_mark = time.perf_counter()
    return hist


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("seed", type=int)
    ap.add_argument("iters", type=int)
    ap.add_argument("out")
    ap.add_argument("--arms", default="engine,engine_td,dropin,ref")
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--lr", type=float, default=5e-3)
    a = ap.parse_args()
    LR0 = a.lr
    student = dict(num_layers=a.layers, hidden_size=a.hidden, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)
    check = [i for i in (250, 500, 1000, 1500, 2000, 3000, 4000, 5000) if i <= a.iters] or [a.iters]
    poses, imgs, train, val = P4.teacher_dataset()
    views = val[:P4.VAL_PER_CHECK]
    res = dict(seed=a.seed, iters=a.iters, lr0=a.lr, student="%dx%d" % (a.layers, a.hidden), rays_per_iter=RAYS, image="%dx%d" % (H, W), arms={})
    for arm in a.arms.split(","):
        res["arms"][arm] = run(arm, a.seed, a.iters, check, student, poses, imgs, train, views)
        json.dump(res, open(a.out, "w"), indent=1)
