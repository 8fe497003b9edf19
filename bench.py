#!/usr/bin/env python
"""bench.py -- NeRF training-iteration throughput on MI355X (BASELINE.json metric: train rays/sec, lego 400x400,
64 coarse + 128 fine samples, 4096 rays/iter, 8x256 nets).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

One "step" = one full training iteration of the reference's loop body (train_nerf.py:210-270) on synthetic data of the
configured shape: select 4096 pixels of a 400x400 view -> generate those rays -> coarse+fine render forward (stratified
+ inverse-CDF sampling, positional encoding, two 8x256 MLPs, compositing) -> MSE loss -> backward through both nets ->
gradient all-reduce over RCCL (N > 1) -> Adam step -> weight re-pack.  fp32 throughout (fp32 MFMA).  Weak scaling:
every rank renders its own 4096 rays per step.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel, from HIP events recorded on the launch stream
inside the timed region; `cpu_baseline` is the oracle (CPU port of the reference path, oracle/nerf_oracle.py) timed on
this box's host cores on a bounded sample (N=1 only).
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import nerf_pytorch_amd as N  # noqa: E402

H = W = 400
FOCAL = 555.5555
RAYS_PER_GPU = 4096
NC, NF = 64, 128
MODEL = dict(num_layers=8, hidden_size=256, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)
FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md chip table


def pose_spherical(theta_deg, phi_deg, radius):
    """Camera-to-world of a camera on a sphere looking at the origin (the 360-degree poses of the blender scenes)."""
    th, ph = math.radians(theta_deg), math.radians(phi_deg)
    t = torch.eye(4)
    t[2, 3] = radius
    rp = torch.tensor([[1, 0, 0, 0], [0, math.cos(ph), -math.sin(ph), 0], [0, math.sin(ph), math.cos(ph), 0], [0, 0, 0, 1.0]])
    rt = torch.tensor([[math.cos(th), 0, -math.sin(th), 0], [0, 1, 0, 0], [math.sin(th), 0, math.cos(th), 0], [0, 0, 0, 1.0]])
    flip = torch.tensor([[-1.0, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]])
    return flip @ rt @ rp @ t


def macs_per_sample(cfg, dx=63, dd=27):
    Wd, L, sk = cfg["hidden_size"], cfg["num_layers"], cfg["skip_connect_every"]
    fwd = dx * Wd
    dgrad = 0
    for i in range(L - 1):
        k = Wd + (dx if (i % sk == 0 and i > 0) else 0)
        fwd += k * Wd
        dgrad += Wd * Wd
    fwd += Wd * Wd + Wd + (Wd + dd) * (Wd // 2) + 3 * (Wd // 2)
    dgrad += Wd * Wd + Wd + Wd * (Wd // 2) + 3 * (Wd // 2)
    return fwd, dgrad


def cpu_baseline(sample_rays=RAYS_PER_GPU):
    """The oracle (kind "port": oracle/nerf_oracle.py, the CPU restatement of the reference path, bit-identical to the
    reference's own functions on CPU -- tests/test_oracle.py) forward + backward on ONE full batch of `sample_rays`
    synthetic rays, 64+128 samples, 8x256 nets, after a 64-ray warm-up (thread pool, allocator)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nerf_oracle as O
    # 16-32 threads is the fastest setting on the 256-thread EPYC host of the GPU box (measured: 8 -> 520, 16 -> 709,
    # 32 -> 576, 64 -> 253, 128 -> 49 rays/s at 128 rays; profiles/r01_cpu_threads.txt): more threads only add
    # fork/join overhead
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    cfg = dict(MODEL)
    pc = {k: v.requires_grad_(True) for k, v in O.init_params(cfg, seed=1).items()}
    pf = {k: v.requires_grad_(True) for k, v in O.init_params(cfg, seed=2).items()}
    g = torch.Generator().manual_seed(0)
    opt = dict(num_coarse=NC, num_fine=NF, perturb=True, lindisp=False, white_background=False, noise_std=0.2)

    def one(n):
        ro = torch.tensor([0.0, 0.0, 4.0]).expand(n, 3)
        rd = torch.randn(n, 3, generator=g) * 0.3
        rd[:, 2] = -1.0
        rays = O.pack_rays(ro, rd, 2.0, 6.0, rd)
        tgt = torch.rand(n, 3, generator=g)
        rand = dict(t_rand=torch.rand(n, NC, generator=g), noise_coarse=torch.randn(n, NC, generator=g),
                    u=torch.rand(n, NF, generator=g), noise_fine=torch.randn(n, NC + NF, generator=g))
        t0 = time.perf_counter()
        out = O.render_rays(rays, pc, pf, cfg, cfg, opt, rand, chunksize=131072)
        loss, _, _, _ = O.loss_and_psnr(out["rgb_coarse"], out["rgb_fine"], tgt)
        loss.backward()
        for p in list(pc.values()) + list(pf.values()):
            p.grad = None
        return time.perf_counter() - t0

    one(64)
    n = sample_rays
    dt = one(n)
    return dict(value=n / dt, unit="rays/s", cores=torch.get_num_threads(), kind="port", seconds=round(dt, 2),
                sample="one full batch of %d rays x (64 coarse + 128 fine), 8x256 nets, fwd+bwd (no optimizer), after a 64-ray "
                       "warm-up; oracle/nerf_oracle.py = the reference's functions restated on torch %s CPU ops (bit-identical "
                       "to the reference on CPU, tests/test_oracle.py)" % (n, torch.__version__))


def dropin_route(dev, n=RAYS_PER_GPU, steps=5, warmup=2):
    """The reference's own loop body on this package's drop-in API: run_one_iter_of_nerf on
    whole-image rays gathered with torch indexing, img2mse, loss.backward(), torch.optim.Adam.step() -- no TrainEngine."""
    cfg = dict(MODEL)
    torch.manual_seed(42)
    mc, mf = N.FlexibleNeRFModel(**cfg).to(dev), N.FlexibleNeRFModel(**cfg).to(dev)
    optim = torch.optim.Adam(list(mc.parameters()) + list(mf.parameters()), lr=5e-3)
    opts = N.make_options(NC, NF, num_random_rays=n)
    ex, ed = N.get_embedding_function(10, True, True), N.get_embedding_function(4, True, True)
    pose = pose_spherical(30.0, -30.0, 4.0).to(dev)
    g = torch.Generator(device=dev).manual_seed(7)
    image = torch.rand(H, W, 3, generator=g, device=dev)

    def step():
        ro, rd = N.get_ray_bundle(H, W, FOCAL, pose)                      # train_nerf.py:213
        sel = torch.randperm(H * W, device=dev)[:n]                        # :218-222 (np.random.choice there)
        ro, rd, tgt = ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel], image.reshape(-1, 3)[sel]
        out = N.run_one_iter_of_nerf(H, W, FOCAL, mc, mf, ro, rd, opts, mode="train", encode_position_fn=ex,
                                     encode_direction_fn=ed)
        loss = N.img2mse(out[0], tgt) + N.img2mse(out[3], tgt)             # :244-258
        loss.backward()
        optim.step()
        optim.zero_grad()
        return loss

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    del mc, mf, optim
    torch.cuda.empty_cache()
    return dict(value=round(n / dt, 1), unit="rays/s", ms_per_step=round(dt * 1e3, 3),
                what="run_one_iter_of_nerf + img2mse + backward + torch.optim.Adam on %d rays (whole-image get_ray_bundle, "
                     "torch randperm gather), %d steps after %d warm-up" % (n, steps, warmup))


def wgrad_bytes_per_sample(cfg, dx_slots=64, dd_slots=32):
    """Algorithmic HBM read bytes of k_wgrad per sample point: every job reads its d(pre-activation) rows and its
    activation rows once (wgrad.hip; rows as laid out by plan.cpp build_layouts_and_jobs)."""
    Wd, L, sk = cfg["hidden_size"], cfg["num_layers"], cfg["skip_connect_every"]
    rows = (Wd + dx_slots)                                   # layer1: dP_0 x X
    for i in range(L - 1):
        rows += 2 * Wd                                        # dP_{i+1} x H_i
        if i % sk == 0 and i > 0:
            rows += Wd + dx_slots                             # ... x X (skip columns)
    rows += 2 * Wd + (32 + Wd) + (Wd // 2 + Wd) + (Wd // 2 + dd_slots) + (32 + Wd // 2)   # feat, alpha, dir, dir x D, rgb
    return 4 * rows


def stash_bytes_per_sample(cfg, dx_slots=64, dd_slots=32):
    """Algorithmic HBM write bytes of the training forward per sample point: the activation stash + ReLU masks."""
    Wd, L = cfg["hidden_size"], cfg["num_layers"]
    return 4 * (dx_slots + dd_slots + L * Wd + Wd + Wd // 2) + 8 * (L + 1)


def pytorch_rocm_reference(dev, n=RAYS_PER_GPU, reps=3):
    """The reference's own PyTorch path (the oracle's torch ops, op for op) on THIS GPU: forward+backward on n rays --
    the denominator of the north star's "x the reference single-GPU PyTorch-ROCm rays/sec"."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nerf_oracle as O
    cfg = dict(MODEL)
    pc = {k: v.to(dev).requires_grad_(True) for k, v in O.init_params(cfg, 1).items()}
    pf = {k: v.to(dev).requires_grad_(True) for k, v in O.init_params(cfg, 2).items()}
    g = torch.Generator().manual_seed(0)
    ro = torch.tensor([0.0, 0.0, 4.0]).expand(n, 3)
    rd = torch.randn(n, 3, generator=g) * 0.3
    rd[:, 2] = -1.0
    rays = O.pack_rays(ro, rd, 2.0, 6.0, rd).to(dev)
    tgt = torch.rand(n, 3, generator=g).to(dev)
    opt = dict(num_coarse=NC, num_fine=NF, perturb=True, lindisp=False, white_background=False, noise_std=0.2)
    best = float("inf")
    for it in range(reps + 1):
        rand = dict(t_rand=torch.rand(n, NC, device=dev), noise_coarse=torch.randn(n, NC, device=dev),
                    u=torch.rand(n, NF, device=dev), noise_fine=torch.randn(n, NC + NF, device=dev))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = O.render_rays(rays, pc, pf, cfg, cfg, opt, rand, chunksize=131072)
        loss = torch.nn.functional.mse_loss(out["rgb_coarse"], tgt) + torch.nn.functional.mse_loss(out["rgb_fine"], tgt)
        loss.backward()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        for p in list(pc.values()) + list(pf.values()):
            p.grad = None
        if it > 0:
            best = min(best, dt)
    del pc, pf, out, loss
    torch.cuda.empty_cache()
    return dict(value=n / best, unit="rays/s", what="oracle torch ops on cuda (== reference PyTorch-ROCm path), "
                "fwd+bwd, no optimizer, %d rays, best of %d" % (n, reps))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rays", type=int, default=RAYS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--hidden", type=int, default=MODEL["hidden_size"])
    ap.add_argument("--layers", type=int, default=MODEL["num_layers"])
    ap.add_argument("--overlap", type=int, default=-1, help="1: two-stream step (coarse backward next to the fine pass); "
                    "0: single-stream order; -1: the engine's default for the net width")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # test hook (scripts/gpu_dp2_smoke.sh): exercise the N > 1 code path on a ONE-GPU box -- every rank on cuda:0 and a
    # gloo process group (RCCL refuses two ranks on one device).  Never set by the driver.
    one_device = os.environ.get("NERFHIP_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_device:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=dev)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    cfg = dict(MODEL, hidden_size=args.hidden, num_layers=args.layers)
    torch.manual_seed(42)  # config/lego.yml:8; every rank builds identical weights
    mc = N.FlexibleNeRFModel(**cfg).to(dev)
    mf = N.FlexibleNeRFModel(**cfg).to(dev)
    eng = N.TrainEngine(mc, mf, NC, NF, perturb=True, lindisp=False, white_background=False, noise_std=0.2, lr=5e-3,
                        seed=1234, world_size=world, rank=rank, overlap=None if args.overlap < 0 else bool(args.overlap))
    n = args.rays
    opts = N.make_options(NC, NF, num_random_rays=n)
    poses = torch.stack([pose_spherical(th, -30.0, 4.0) for th in torch.linspace(-180, 180, 101)[:-1].tolist()]).to(dev)
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    images = torch.rand(8, H, W, 3, generator=g, device=dev)        # synthetic training views, resident in HBM
    lib = N._lib.get_lib()

    def one_step(i):
        # the reference's loop body, train_nerf.py:210-270: pick a view, draw 4096 distinct pixels, their rays and
        # targets (one launch, on the device), forward, loss, backward, [all-reduce], Adam with the decayed lr
        k = i * world + rank
        return eng.step_on_image(images[k % 8], poses[k % poses.shape[0]], H, W, FOCAL, opts, n, lr=N.TrainEngine.lr_at(i))

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        one_step(i)
    fence()
    lib.profile_enable(1)
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        loss = one_step(i)
    fence()
    dt = time.perf_counter() - t0
    lib.profile_enable(0)
    import ctypes
    buf = ctypes.create_string_buffer(1 << 16)
    lib.profile_report(buf, len(buf))
    tmax = torch.tensor([dt], device=dev)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax)
    loss_host = [float(v) for v in loss.cpu()]

    if rank == 0:
        kern = {}
        for line in buf.value.decode().splitlines():
            name, cnt, ms = line.rsplit(" ", 2)
            kern[name] = (int(cnt), float(ms))
        fwd_macs, dgrad_macs = macs_per_sample(cfg)
        m_c, m_f = n * NC, n * (NC + NF)
        # algorithmic FLOPs per launch of each MLP kernel: 2 * MACs/sample * samples of the launch
        flops = {"fwd": 2.0 * fwd_macs, "dgrad": 2.0 * dgrad_macs, "wgrad": 2.0 * fwd_macs}
        table = []
        for name, (cnt, ms) in kern.items():
            kind = "fwd" if "k_mlp_fwd" in name else "dgrad" if "k_mlp_dgrad" in name else "wgrad" if "k_wgrad" in name and "reduce" not in name else None
            table.append((ms, name, cnt, kind))
        table.sort(reverse=True)
        dom = next((t for t in table if t[3] is not None), None)
        roof = None
        if dom is not None:
            ms, name, cnt, kind = dom
            # the kernel is launched once per net per step: coarse (m_c samples) and fine (m_f samples)
            per_step_flops = flops[kind] * (m_c + m_f)
            launches_per_step = cnt / args.steps
            avg_ms = ms / cnt
            achieved = per_step_flops / launches_per_step / (avg_ms * 1e-3) / 1e12
            # HBM traffic per launch of that kernel (mean of the coarse- and the fine-sized launch): algorithmic bytes,
            # and the counter bytes of the tracked rocprofv3 --pmc pass of this same command (scripts/gpu_pmc.sh:
            # separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE x2 as MI355X_MICROARCH.md prescribes for gfx950)
            samples_per_launch = (m_c + m_f) / 2.0
            alg = {"wgrad": wgrad_bytes_per_sample(cfg), "fwd": stash_bytes_per_sample(cfg),
                   "dgrad": 4 * (cfg["num_layers"] * cfg["hidden_size"] + cfg["hidden_size"] + cfg["hidden_size"] // 2 + 32)}[kind]
            traffic = dict(algorithmic_gb=round(alg * samples_per_launch / 1e9, 3), counter_gb=None, source=None)
            pmc_file = os.path.join(ROOT, "profiles", "r02_pmc_summary.json")
            if os.path.exists(pmc_file) and cfg == MODEL and n == RAYS_PER_GPU:
                pm = json.load(open(pmc_file))
                rows = [v for v in pm.values() if v["kernel"] == ("k_wgrad" if kind == "wgrad" else "k_mlp_%s16" % kind)]
                if rows:
                    per = [(r["fetch_gb_x2"] if kind == "wgrad" else 0.0) + (r["write_gb"] if kind != "wgrad" else 0.0) for r in rows]
                    traffic["counter_gb"] = round(sum(per) / len(per), 3)
                    traffic["source"] = "profiles/r02_pmc_summary.json (tracked rocprofv3 --pmc passes of this command: " + \
                        ("FETCH_SIZE x2" if kind == "wgrad" else "WRITE_SIZE") + ", mean over the launch sizes recorded)"
            roof = dict(bound="mfma", kernel=name.strip("()"), achieved=round(achieved, 3), peak=FP32_MFMA_PEAK_TFLOPS,
                        unit="TFLOP/s", frac=round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), traffic=traffic,
                        avg_launch_ms=round(avg_ms, 4), launches=cnt,
                        algorithmic_gflop_per_launch=round(per_step_flops / launches_per_step / 1e9, 2),
                        kernel_ms_per_step={nm.strip("()"): round(m / args.steps, 4) for m, nm, _, _ in table})
        total_flops = (2.0 * (2 * fwd_macs + dgrad_macs)) * (m_c + m_f)
        res = dict(metric="train rays/sec", value=round(world * n * args.steps / dt, 2), unit="rays/s", n_gpus=world,
                   steps=args.steps, warmup=args.warmup, ms_per_step=round(dt / args.steps * 1e3, 3),
                   higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
                   config=dict(workload="lego 400x400 synthetic views (BASELINE configs[1]): %d rays/GPU/iter, %d coarse + "
                                        "%d fine samples, %dx%d coarse+fine nets, perturb, noise 0.2, Adam, full iteration"
                                        % (n, NC, NF, cfg["num_layers"], cfg["hidden_size"]),
                               rays_per_gpu=n, global_rays=n * world, parallelism="dp%d" % world,
                               two_stream_step=bool(eng.overlap)),
                   step_tflops=round(total_flops / (dt / args.steps) / 1e12, 2),
                   step_frac_of_fp32_mfma_peak=round(total_flops / (dt / args.steps) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                   final_loss=loss_host, roofline=roof)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
            try:
                res["dropin_route"] = dropin_route(dev)
            except Exception as e:
                res["dropin_route"] = dict(error=repr(e)[:200])
            try:
                res["pytorch_rocm_reference"] = pytorch_rocm_reference(dev)
                res["speedup_vs_pytorch_rocm_fwd_bwd"] = round(res["value"] / res["pytorch_rocm_reference"]["value"], 3)
            except Exception as e:  # the torch arm needs ~13 GB and must never take the bench line down
                res["pytorch_rocm_reference"] = dict(error=repr(e)[:200])
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
