"""GPU-box experiment: PSNR@iters, HIP path vs the reference's PyTorch path (oracle torch ops on the same GPU).

No dataset is available offline, so the scene is a *teacher*: the reference's pretrained lego-lowres nets
(tests/golden/lego_lowres_weights.npz, 4x128) rendered at 100x100 from poses on the blender 360-degree sphere.  Two
students of the north-star geometry (8x256, skip 4) with identical init (torch.manual_seed(42)) and identical data
order are trained on it:  arm A = reference PyTorch ops + torch.optim.Adam,  arm B = TrainEngine (fused HIP step).
PSNR is the reference's definition, -10 log10(coarse_mse + fine_mse) (train_nerf.py:258-260), on the training batch
(mean of the last 25 iterations) and on a held-out view (deterministic render).
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
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
RAYS = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
CHECK = [i for i in (100, 250, 500, 1000, 1500, 2500, 4000) if i <= ITERS]
H = W = 100
FOCAL = 0.5 * W / math.tan(0.5 * 0.6911112070083618)
NC, NF = 64, 128
STUDENT = dict(num_layers=8, hidden_size=256, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)


def pose_spherical(theta_deg, phi_deg, radius):
    th, ph = math.radians(theta_deg), math.radians(phi_deg)
    t = torch.eye(4)
    t[2, 3] = radius
    rp = torch.tensor([[1, 0, 0, 0], [0, math.cos(ph), -math.sin(ph), 0], [0, math.sin(ph), math.cos(ph), 0], [0, 0, 0, 1.0]])
    rt = torch.tensor([[math.cos(th), 0, -math.sin(th), 0], [0, 1, 0, 0], [math.sin(th), 0, math.cos(th), 0], [0, 0, 0, 1.0]])
    flip = torch.tensor([[-1.0, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]])
    return flip @ rt @ rp @ t


def teacher_dataset(n_train=40, n_val=4):
    w = np.load(os.path.join(ROOT, "tests", "golden", "lego_lowres_weights.npz"))
    cfg = dict(num_layers=4, hidden_size=128, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)
    mc, mf = N.FlexibleNeRFModel(**cfg), N.FlexibleNeRFModel(**cfg)
    mc.load_state_dict({k[2:]: torch.from_numpy(w[k]) for k in w.files if k.startswith("c_")})
    mf.load_state_dict({k[2:]: torch.from_numpy(w[k]) for k in w.files if k.startswith("f_")})
    mc, mf = mc.to(dev), mf.to(dev)
    opts = N.make_options(64, 64, perturb=False, white_background=True, radiance_field_noise_std=0.0)
    ex, ed = N.get_embedding_function(10, True, True), N.get_embedding_function(4, True, True)
    thetas = torch.linspace(-180, 180, n_train + n_val + 1)[:-1].tolist()
    poses, imgs = [], []
    with torch.no_grad():
        for th in thetas:
            pose = pose_spherical(th, -30.0, 4.0).to(dev)
            ro, rd = N.get_ray_bundle(H, W, FOCAL, pose[:3, :4])
            out = N.run_one_iter_of_nerf(H, W, FOCAL, mc, mf, ro, rd, opts, mode="validation", encode_position_fn=ex,
                                         encode_direction_fn=ed)
            poses.append(pose)
            imgs.append(out[3].clamp(0, 1))
    poses, imgs = torch.stack(poses), torch.stack(imgs)
    val = list(range(0, n_train + n_val, (n_train + n_val) // n_val))[:n_val]
    train = [i for i in range(n_train + n_val) if i not in val]
    return poses, imgs, train, val


def psnr(v):
    return -10.0 * math.log10(v if v > 0 else 1e-5)


def data_stream(poses, imgs, train, seed=7):
    g = torch.Generator().manual_seed(seed)
    while True:
        v = train[int(torch.randint(len(train), (1,), generator=g))]
        pix = torch.randperm(H * W, generator=g)[:RAYS].to(dev)
        ro, rd = N.get_rays_at_pixels(H, W, FOCAL, poses[v][:3, :4], pix)
        yield ro, rd, imgs[v].reshape(-1, 3)[pix].contiguous()


OPT = dict(num_coarse=NC, num_fine=NF, perturb=True, lindisp=False, white_background=True, noise_std=0.2)
EVAL = dict(num_coarse=NC, num_fine=NF, perturb=False, lindisp=False, white_background=True, noise_std=0.0)


def run_reference_arm(poses, imgs, train, val):
    torch.manual_seed(42)
    mc, mf = N.FlexibleNeRFModel(**STUDENT), N.FlexibleNeRFModel(**STUDENT)   # only for the identical init
    pc = {k: v.detach().clone().to(dev).requires_grad_(True) for k, v in mc.state_dict().items()}
    pf = {k: v.detach().clone().to(dev).requires_grad_(True) for k, v in mf.state_dict().items()}
    opt = torch.optim.Adam(list(pc.values()) + list(pf.values()), lr=5e-3)
    stream = data_stream(poses, imgs, train)
    hist, recent, t0 = {}, [], time.perf_counter()
    for i in range(1, ITERS + 1):
        ro, rd, tgt = next(stream)
        rays = O.pack_rays(ro, rd, 2.0, 6.0, rd)
        n = rays.shape[0]
        rand = dict(t_rand=torch.rand(n, NC, device=dev), noise_coarse=torch.randn(n, NC, device=dev),
                    u=torch.rand(n, NF, device=dev), noise_fine=torch.randn(n, NC + NF, device=dev))
        out = O.render_rays(rays, pc, pf, STUDENT, STUDENT, OPT, rand, chunksize=131072)
        loss = torch.nn.functional.mse_loss(out["rgb_coarse"], tgt) + torch.nn.functional.mse_loss(out["rgb_fine"], tgt)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        for gq in opt.param_groups:
            gq["lr"] = N.TrainEngine.lr_at(i)
        recent.append(float(loss))
        recent = recent[-25:]
        if i in CHECK:
            with torch.no_grad():
                vals = []
                for v in val:
                    ro_i, rd_i = N.get_ray_bundle(H, W, FOCAL, poses[v][:3, :4])
                    r = O.pack_rays(ro_i.reshape(-1, 3), rd_i.reshape(-1, 3), 2.0, 6.0, rd_i.reshape(-1, 3))
                    o = O.render_rays(r, pc, pf, STUDENT, STUDENT, EVAL, chunksize=131072)
                    t = imgs[v].reshape(-1, 3)
                    vals.append((float(torch.mean((o["rgb_coarse"] - t) ** 2)), float(torch.mean((o["rgb_fine"] - t) ** 2))))
            vc, vf = np.mean([a for a, _ in vals]), np.mean([b for _, b in vals])
            torch.cuda.synchronize()
            hist[i] = dict(train_psnr=psnr(float(np.mean(recent))), val_psnr=psnr(vc + vf), val_psnr_fine=psnr(vf),
                           wall_s=time.perf_counter() - t0)
            print("reference-arm", i, hist[i], flush=True)
    return hist


def run_hip_arm(poses, imgs, train, val):
    torch.manual_seed(42)
    mc, mf = N.FlexibleNeRFModel(**STUDENT).to(dev), N.FlexibleNeRFModel(**STUDENT).to(dev)
    eng = N.TrainEngine(mc, mf, NC, NF, perturb=True, white_background=True, noise_std=0.2, lr=5e-3, seed=99)
    opts = N.make_options(NC, NF, white_background=True)
    ev = N.make_options(NC, NF, perturb=False, white_background=True, radiance_field_noise_std=0.0)
    ex, ed = N.get_embedding_function(10, True, True), N.get_embedding_function(4, True, True)
    stream = data_stream(poses, imgs, train)
    hist, recent, t0 = {}, [], time.perf_counter()
    losses = []
    for i in range(1, ITERS + 1):
        ro, rd, tgt = next(stream)
        loss3 = eng.step(N.pack_rays(ro, rd, opts), tgt, lr=N.TrainEngine.lr_at(i - 1))
        losses.append(loss3[2:3].clone())          # stays on the device: no host sync in the loop
        if i in CHECK:
            recent = torch.cat(losses[-25:]).cpu().numpy()
            with torch.no_grad():
                vals = []
                for v in val:
                    ro_i, rd_i = N.get_ray_bundle(H, W, FOCAL, poses[v][:3, :4])
                    o = N.run_one_iter_of_nerf(H, W, FOCAL, mc, mf, ro_i, rd_i, ev, mode="validation",
                                               encode_position_fn=ex, encode_direction_fn=ed)
                    t = imgs[v]
                    vals.append((float(torch.mean((o[0] - t) ** 2)), float(torch.mean((o[3] - t) ** 2))))
            vc, vf = np.mean([a for a, _ in vals]), np.mean([b for _, b in vals])
            torch.cuda.synchronize()
            hist[i] = dict(train_psnr=psnr(float(np.mean(recent))), val_psnr=psnr(vc + vf), val_psnr_fine=psnr(vf),
                           wall_s=time.perf_counter() - t0)
            print("hip-arm", i, hist[i], flush=True)
    return hist


if __name__ == "__main__":
    poses, imgs, train, val = teacher_dataset()
    print("teacher: %d views, mean %.4f" % (imgs.shape[0], float(imgs.mean())), flush=True)
    hb = run_hip_arm(poses, imgs, train, val)
    ha = run_reference_arm(poses, imgs, train, val)
    print(json.dumps(dict(iters=ITERS, rays_per_iter=RAYS, reference_pytorch_rocm=ha, hip=hb)))
