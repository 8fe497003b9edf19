"""GPU-box experiment (round 2): PSNR@iters at the metric's scale -- lego-like scene at 400x400, 64 coarse + 128 fine,
4096 rays/iter, 8x256 nets (BASELINE.json metric: "train rays/sec + PSNR@iters, lego 400x400 64c+128f").

No dataset is available offline, so the scene is a *teacher*: the reference's pretrained lego-lowres nets
(tests/golden/lego_lowres_weights.npz, 4x128) rendered at 400x400 from poses on the blender 360-degree sphere
(100 training views, 10 held-out validation views).  Students: the north-star geometry (8x256, skip 4), identical init
per seed (torch.manual_seed(seed), config/lego.yml:8), identical data order per seed, lr = 5e-3 * 0.1^(i / 250000)
(train_nerf.py:264-270), Adam defaults.

    arm "ref"    : the reference's own PyTorch path on this GPU (oracle torch ops, op for op) + torch.optim.Adam
    arm "dropin" : this package behind the reference API -- run_one_iter_of_nerf + img2mse + backward + torch.optim.Adam;
                   it draws t_rand / noise / u with torch.rand(n) in the reference's order, so with the same torch seed
                   it consumes EXACTLY the random numbers arm "ref" consumes: the arms differ only in kernel arithmetic
    arm "engine" : TrainEngine (fused step, in-kernel Philox draws, fused Adam) -- what bench.py times

PSNR = -10 log10(coarse_mse + fine_mse) (train_nerf.py:258-260): on the training batch (mean loss of the last 50
iterations) and by the validation-image protocol of train_nerf.py:336-347 (whole 400x400 held-out views rendered with
the validation options: perturb off, noise 0), at fixed iteration counts.

    python scripts/psnr400.py ARM SEED ITERS OUT.json [DRAW_SEED]
(DRAW_SEED: other random draws for the same init / data order -- how much of an arm-to-arm difference is the draws alone)
"""
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import nerf_oracle as O  # noqa: E402
import nerf_pytorch_amd as N  # noqa: E402

dev = torch.device("cuda", 0)
H = W = 400
FOCAL = 0.5 * W / math.tan(0.5 * 0.6911112070083618)
NC, NF, RAYS = 64, 128, 4096
N_TRAIN, N_VAL, VAL_PER_CHECK = 100, 10, 3
STUDENT = dict(num_layers=8, hidden_size=256, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)
CACHE = "/tmp/psnr400_teacher.pt"


def pose_spherical(theta_deg, phi_deg, radius):
    th, ph = math.radians(theta_deg), math.radians(phi_deg)
    t = torch.eye(4)
    t[2, 3] = radius
    rp = torch.tensor([[1, 0, 0, 0], [0, math.cos(ph), -math.sin(ph), 0], [0, math.sin(ph), math.cos(ph), 0], [0, 0, 0, 1.0]])
    rt = torch.tensor([[math.cos(th), 0, -math.sin(th), 0], [0, 1, 0, 0], [math.sin(th), 0, math.cos(th), 0], [0, 0, 0, 1.0]])
    flip = torch.tensor([[-1.0, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]])
    return flip @ rt @ rp @ t


def teacher_dataset():
    if os.path.exists(CACHE):
        d = torch.load(CACHE)
        return d["poses"].to(dev), d["imgs"].to(dev), d["train"], d["val"]
    w = np.load(os.path.join(ROOT, "tests", "golden", "lego_lowres_weights.npz"))
    cfg = dict(num_layers=4, hidden_size=128, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)
    mc, mf = N.FlexibleNeRFModel(**cfg), N.FlexibleNeRFModel(**cfg)
    mc.load_state_dict({k[2:]: torch.from_numpy(w[k]) for k in w.files if k.startswith("c_")})
    mf.load_state_dict({k[2:]: torch.from_numpy(w[k]) for k in w.files if k.startswith("f_")})
    mc, mf = mc.to(dev), mf.to(dev)
    opts = N.make_options(64, 64, perturb=False, white_background=True, radiance_field_noise_std=0.0)
    ex, ed = N.get_embedding_function(10, True, True), N.get_embedding_function(4, True, True)
    n = N_TRAIN + N_VAL
    g = torch.Generator().manual_seed(2020)
    thetas = (torch.linspace(-180, 180, n + 1)[:-1] + 1.3 * torch.rand(n, generator=g)).tolist()
    phis = (-50.0 + 40.0 * torch.rand(n, generator=g)).tolist()  # elevations between -50 and -10 degrees
    poses, imgs = [], []
    with torch.no_grad():
        for th, ph in zip(thetas, phis):
            pose = pose_spherical(th, ph, 4.0).to(dev)
            ro, rd = N.get_ray_bundle(H, W, FOCAL, pose[:3, :4])
            out = N.run_one_iter_of_nerf(H, W, FOCAL, mc, mf, ro, rd, opts, mode="validation", encode_position_fn=ex,
                                         encode_direction_fn=ed)
            poses.append(pose)
            imgs.append(out[3].clamp(0, 1))
    poses, imgs = torch.stack(poses), torch.stack(imgs)
    val = list(range(5, n, n // N_VAL))[:N_VAL]
    train = [i for i in range(n) if i not in val]
    torch.save(dict(poses=poses.cpu(), imgs=imgs.cpu(), train=train, val=val), CACHE)
    return poses, imgs, train, val


def psnr(v):
    return -10.0 * math.log10(v if v > 0 else 1e-5)


def data_stream(poses, imgs, train, seed):
    """train_nerf.py:203-227: a random training view, 4096 distinct random pixels of it (host generator: identical
    order in every arm)."""
    g = torch.Generator().manual_seed(1000 + seed)
    while True:
        v = train[int(torch.randint(len(train), (1,), generator=g))]
        pix = torch.randperm(H * W, generator=g)[:RAYS].to(dev)
        ro, rd = N.get_rays_at_pixels(H, W, FOCAL, poses[v][:3, :4], pix)
        yield ro, rd, imgs[v].reshape(-1, 3)[pix].contiguous()


OPT = dict(num_coarse=NC, num_fine=NF, perturb=True, lindisp=False, white_background=True, noise_std=0.2)
EVAL = dict(num_coarse=NC, num_fine=NF, perturb=False, lindisp=False, white_background=True, noise_std=0.0)


def validate_ref(pc, pf, poses, imgs, views):
    vals = []
    with torch.no_grad():
        for v in views:
            ro_i, rd_i = N.get_ray_bundle(H, W, FOCAL, poses[v][:3, :4])
            r = O.pack_rays(ro_i.reshape(-1, 3), rd_i.reshape(-1, 3), 2.0, 6.0, rd_i.reshape(-1, 3))
            mc_, mf_ = [], []
            for lo in range(0, r.shape[0], 32768):  # the torch path materialises (rays x 192 x 256) activations: chunk the image
                o = O.render_rays(r[lo:lo + 32768], pc, pf, STUDENT, STUDENT, EVAL, chunksize=131072)
                t = imgs[v].reshape(-1, 3)[lo:lo + 32768]
                mc_.append(((o["rgb_coarse"] - t) ** 2).sum())
                mf_.append(((o["rgb_fine"] - t) ** 2).sum())
            vals.append((float(sum(mc_)) / (3 * r.shape[0]), float(sum(mf_)) / (3 * r.shape[0])))
    return vals


def validate_hip(mc, mf, poses, imgs, views):
    ev = N.make_options(NC, NF, perturb=False, white_background=True, radiance_field_noise_std=0.0)
    ex, ed = N.get_embedding_function(10, True, True), N.get_embedding_function(4, True, True)
    vals = []
    with torch.no_grad():
        for v in views:
            ro_i, rd_i = N.get_ray_bundle(H, W, FOCAL, poses[v][:3, :4])
            o = N.run_one_iter_of_nerf(H, W, FOCAL, mc, mf, ro_i, rd_i, ev, mode="validation", encode_position_fn=ex,
                                       encode_direction_fn=ed)
            t = imgs[v]
            vals.append((float(torch.mean((o[0] - t) ** 2)), float(torch.mean((o[3] - t) ** 2))))
    return vals


def record(hist, i, recent, vals, t_train, t0):
    vc, vf = float(np.mean([a for a, _ in vals])), float(np.mean([b for _, b in vals]))
    hist[i] = dict(train_psnr=psnr(float(np.mean(recent))), val_psnr=psnr(vc + vf), val_psnr_fine=psnr(vf),
                   train_wall_s=round(t_train, 2), wall_s=round(time.perf_counter() - t0, 2))
    print(i, hist[i], flush=True)


def run(arm, seed, iters, check, draw_seed=None):
    poses, imgs, train, val = teacher_dataset()
    views = val[:VAL_PER_CHECK]
    torch.manual_seed(seed)
    mc, mf = N.FlexibleNeRFModel(**STUDENT), N.FlexibleNeRFModel(**STUDENT)   # nn.Linear default init, reference order
    stream = data_stream(poses, imgs, train, seed)
    hist, losses = {}, []
    t0 = time.perf_counter()
    t_train, t_mark = 0.0, time.perf_counter()
    if arm == "ref":
        pc = {k: v.detach().clone().to(dev).requires_grad_(True) for k, v in mc.state_dict().items()}
        pf = {k: v.detach().clone().to(dev).requires_grad_(True) for k, v in mf.state_dict().items()}
        opt = torch.optim.Adam(list(pc.values()) + list(pf.values()), lr=5e-3)
    elif arm == "dropin":
        mc, mf = mc.to(dev), mf.to(dev)
        opt = torch.optim.Adam(list(mc.parameters()) + list(mf.parameters()), lr=5e-3)
        opts = N.make_options(NC, NF, white_background=True)
        ex, ed = N.get_embedding_function(10, True, True), N.get_embedding_function(4, True, True)
    else:
        mc, mf = mc.to(dev), mf.to(dev)
        eng = N.TrainEngine(mc, mf, NC, NF, perturb=True, white_background=True, noise_std=0.2, lr=5e-3, seed=seed)
        opts = N.make_options(NC, NF, white_background=True)
    torch.manual_seed(seed + 12345 if draw_seed is None else draw_seed)  # the draws of the training loop (arms "ref" and "dropin" consume the same numbers)
    for i in range(1, iters + 1):
        ro, rd, tgt = next(stream)
        if arm == "ref":
            rays = O.pack_rays(ro, rd, 2.0, 6.0, rd)
            n = rays.shape[0]
            rand = dict(t_rand=torch.rand((n, NC), dtype=torch.float32, device=dev),
                        noise_coarse=torch.randn((n, NC), dtype=torch.float32, device=dev),
                        u=torch.rand((n, NF), dtype=torch.float32, device=dev),
                        noise_fine=torch.randn((n, NC + NF), dtype=torch.float32, device=dev))
            out = O.render_rays(rays, pc, pf, STUDENT, STUDENT, OPT, rand, chunksize=131072)
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
            loss3 = eng.step(N.pack_rays(ro, rd, opts), tgt, lr=N.TrainEngine.lr_at(i - 1))
            losses.append(loss3[2:3].clone())
        if arm != "engine":
            for gq in opt.param_groups:  # train_nerf.py:264-270
                gq["lr"] = N.TrainEngine.lr_at(i)
        if i in check:
            torch.cuda.synchronize()
            t_train += time.perf_counter() - t_mark
            recent = torch.cat(losses[-50:]).cpu().numpy()
            vals = validate_ref(pc, pf, poses, imgs, views) if arm == "ref" else validate_hip(mc, mf, poses, imgs, views)
            record(hist, i, recent, vals, t_train, t0)
            losses = losses[-50:]
            t_mark = time.perf_counter()
    return hist


if __name__ == "__main__":
    arm, seed, iters, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    check = [i for i in (250, 500, 1000, 2000, 3000, 4000, 5000, 7500, 10000) if i <= iters]
    check = check or [iters]
    draw_seed = int(sys.argv[5]) if len(sys.argv) > 5 else None
    hist = run(arm, seed, iters, check, draw_seed)
    json.dump(dict(arm=arm, seed=seed, draw_seed=draw_seed, iters=iters, rays_per_iter=RAYS, image="%dx%d" % (H, W), train_views=N_TRAIN,
                   val_views_per_check=VAL_PER_CHECK, checkpoints=hist), open(out, "w"), indent=1)
