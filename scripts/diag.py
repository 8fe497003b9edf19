"""GPU-box diagnostics (not a test): error magnitudes of the HIP path vs the oracle on CPU, with the oracle's torch
ops run on the GPU (== the reference's PyTorch-ROCm path, op for op) as the fp32 noise-floor yardstick; PyTorch-ROCm
reference throughput; CPU-baseline thread sweep.  Writes JSON lines."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import nerf_oracle as O  # noqa: E402
import nerf_pytorch_amd as N  # noqa: E402

dev = torch.device("cuda", 0)
CFG = dict(num_layers=8, hidden_size=256, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)


def stats(a, b):
    e = np.abs(np.nan_to_num(a) - np.nan_to_num(b)).ravel()
    return dict(max=float(e.max()), p999=float(np.quantile(e, 0.999)), p50=float(np.median(e)))


def emit(**kw):
    print(json.dumps(kw), flush=True)


def errors(n=256, nc=64, nf=128, noise=0.3):
    g = torch.Generator().manual_seed(5)
    pc, pf = O.init_params(CFG, 6), O.init_params(CFG, 7)
    ro = torch.tensor([0.2, -0.1, 4.0]).expand(n, 3) + 0.05 * torch.randn(n, 3, generator=g)
    rd = torch.randn(n, 3, generator=g) * 0.3
    rd[:, 2] = -1.0
    rays = O.pack_rays(ro, rd, 2.0, 6.0, rd)
    rand = dict(t_rand=torch.rand(n, nc, generator=g), noise_coarse=torch.randn(n, nc, generator=g),
                u=torch.rand(n, nf, generator=g), noise_fine=torch.randn(n, nc + nf, generator=g))
    opt = dict(num_coarse=nc, num_fine=nf, perturb=True, lindisp=False, white_background=False, noise_std=noise)
    with torch.no_grad():
        cpu = O.render_rays(rays, pc, pf, CFG, CFG, opt, rand)
        tg = O.render_rays(rays.to(dev), {k: v.to(dev) for k, v in pc.items()}, {k: v.to(dev) for k, v in pf.items()}, CFG,
                           CFG, opt, {k: v.to(dev) for k, v in rand.items()})
    mc, mf = N.FlexibleNeRFModel(**CFG), N.FlexibleNeRFModel(**CFG)
    mc.load_state_dict(pc)
    mf.load_state_dict(pf)
    mc, mf = mc.to(dev), mf.to(dev)
    # unit MLP on the oracle's own encoded coarse points
    pts = (rays[:, None, :3] + rays[:, None, 3:6] * cpu["z_coarse"][..., None]).reshape(-1, 3)
    emb = torch.cat([O.positional_encoding(pts, 10), O.positional_encoding(rays[:, None, -3:].expand(n, nc, 3).reshape(-1, 3), 4)], -1)
    with torch.no_grad():
        raw_cpu = O.mlp_forward(pc, emb, CFG)
        raw_hip = mc(emb.to(dev)).cpu()
        raw_tg = O.mlp_forward({k: v.to(dev) for k, v in pc.items()}, emb.to(dev), CFG).cpu()
    emit(what="mlp_raw_vs_cpu", hip=stats(raw_hip.numpy(), raw_cpu.numpy()), torch_rocm=stats(raw_tg.numpy(), raw_cpu.numpy()),
         raw_absmax=float(raw_cpu.abs().max()))
    opts = N.make_options(nc, nf, perturb=True, radiance_field_noise_std=noise)
    ex, ed = N.get_embedding_function(10, True, True), N.get_embedding_function(4, True, True)
    q = [rand[k].to(dev) for k in ("t_rand", "noise_coarse", "u", "noise_fine")]
    real = torch.rand, torch.randn
    torch.rand = lambda *a, **k: q.pop(0)
    torch.randn = lambda *a, **k: q.pop(0)
    try:
        with torch.no_grad():
            hip = N.predict_and_render_radiance(rays.to(dev), mc, mf, opts, encode_position_fn=ex, encode_direction_fn=ed)
    finally:
        torch.rand, torch.randn = real
    for i, k in enumerate(("rgb_coarse", "disp_coarse", "acc_coarse", "rgb_fine", "disp_fine", "acc_fine")):
        emit(what="render_vs_cpu", out=k, hip=stats(hip[i].cpu().numpy(), cpu[k].numpy()),
             torch_rocm=stats(tg[k].cpu().numpy(), cpu[k].numpy()))


def torch_rocm_reference(n=4096, nc=64, nf=128, reps=3):
    """The reference's PyTorch path (oracle ops) on the GPU: forward+backward, like BASELINE.md section 3 item 2."""
    g = torch.Generator().manual_seed(0)
    pc = {k: v.to(dev).requires_grad_(True) for k, v in O.init_params(CFG, 1).items()}
    pf = {k: v.to(dev).requires_grad_(True) for k, v in O.init_params(CFG, 2).items()}
    ro = torch.tensor([0.0, 0.0, 4.0]).expand(n, 3)
    rd = torch.randn(n, 3, generator=g) * 0.3
    rd[:, 2] = -1.0
    rays = O.pack_rays(ro, rd, 2.0, 6.0, rd).to(dev)
    tgt = torch.rand(n, 3, generator=g).to(dev)
    opt = dict(num_coarse=nc, num_fine=nf, perturb=True, lindisp=False, white_background=False, noise_std=0.2)
    times, times_f = [], []
    for it in range(reps + 1):
        rand = dict(t_rand=torch.rand(n, nc, device=dev), noise_coarse=torch.randn(n, nc, device=dev),
                    u=torch.rand(n, nf, device=dev), noise_fine=torch.randn(n, nc + nf, device=dev))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = O.render_rays(rays, pc, pf, CFG, CFG, opt, rand, chunksize=131072)
        loss = torch.nn.functional.mse_loss(out["rgb_coarse"], tgt) + torch.nn.functional.mse_loss(out["rgb_fine"], tgt)
        loss.backward()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        for p in list(pc.values()) + list(pf.values()):
            p.grad = None
        with torch.no_grad():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            O.render_rays(rays, pc, pf, CFG, CFG, opt, rand, chunksize=131072)
            torch.cuda.synchronize()
            dtf = time.perf_counter() - t0
        if it > 0:
            times.append(dt)
            times_f.append(dtf)
    emit(what="torch_rocm_reference", n=n, fwd_bwd_rays_per_s=n / min(times), fwd_rays_per_s=n / min(times_f),
         peak_mem_GB=torch.cuda.max_memory_allocated() / 2 ** 30)


def cpu_sweep(n=128, nc=64, nf=128):
    g = torch.Generator().manual_seed(0)
    ro = torch.tensor([0.0, 0.0, 4.0]).expand(n, 3)
    rd = torch.randn(n, 3, generator=g) * 0.3
    rd[:, 2] = -1.0
    rays = O.pack_rays(ro, rd, 2.0, 6.0, rd)
    tgt = torch.rand(n, 3, generator=g)
    opt = dict(num_coarse=nc, num_fine=nf, perturb=True, lindisp=False, white_background=False, noise_std=0.2)
    rand = dict(t_rand=torch.rand(n, nc), noise_coarse=torch.randn(n, nc), u=torch.rand(n, nf), noise_fine=torch.randn(n, nc + nf))
    for th in (8, 16, 32, 64, 128):
        torch.set_num_threads(th)
        pc = {k: v.requires_grad_(True) for k, v in O.init_params(CFG, 1).items()}
        pf = {k: v.requires_grad_(True) for k, v in O.init_params(CFG, 2).items()}
        best = 1e9
        for it in range(3):
            t0 = time.perf_counter()
            out = O.render_rays(rays, pc, pf, CFG, CFG, opt, rand)
            loss, _, _, _ = O.loss_and_psnr(out["rgb_coarse"], out["rgb_fine"], tgt)
            loss.backward()
            dt = time.perf_counter() - t0
            if it > 0:
                best = min(best, dt)
        emit(what="cpu_threads", threads=th, rays_per_s=n / best)


if __name__ == "__main__":
    which = sys.argv[1:] or ["errors", "torch", "cpu"]
    if "errors" in which:
        errors()
    if "torch" in which:
        torch_rocm_reference()
    if "cpu" in which:
        cpu_sweep()
