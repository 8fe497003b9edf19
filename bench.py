#!/usr/bin/env python
"""bench.py -- NeRF training-iteration throughput on MI355X (BASELINE.json metric: train rays/sec, lego 400x400,
64 coarse + 128 fine samples, 4096 rays/iter, 8x256 nets).

    python bench.py --gpus N --steps K --warmup W
    N > 1: either launched by `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` (one rank per GPU, RCCL), or invoked exactly as above -- bench.py then
    re-launches itself that way.
    --global-rays G     strong scaling: the step's G rays are sharded over the ranks (BASELINE configs[2]:
                        `--image 800 --global-rays 8192` = 1024 rays/GPU at N = 8); default is weak scaling
                        (every rank renders its own --rays = 4096)
    --mode eval         BASELINE configs[4]: inference-only 360-degree render, 800x800 poses, rows of every pose
                        sharded over the ranks (no collective); one "step" = one pose

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
import socket
import subprocess
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
FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md chip table (256 CUs x 256 FLOP/clk at 2.4 GHz)
BF16X3_PEAK_TFLOPS = 2500.0 / 3.0  # dense bf16 MFMA peak (same table) / three MFMAs per fp32-equivalent product block
HBM_PEAK_TBS = 8.0             # same guide: HBM3E ~8 TB/s
PEAK_CLOCK_GHZ = 2.4


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


def cpu_baseline_eval(sample_rays=8192):
    """Config 5's CPU leg: the oracle's forward-only render (no_grad, perturb off, noise 0) of `sample_rays` rays."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nerf_oracle as O
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    cfg = dict(MODEL)
    pc, pf = O.init_params(cfg, seed=1), O.init_params(cfg, seed=2)
    g = torch.Generator().manual_seed(0)
    opt = dict(num_coarse=NC, num_fine=NF, perturb=False, lindisp=False, white_background=False, noise_std=0.0)

    def one(n):
        ro = torch.tensor([0.0, 0.0, 4.0]).expand(n, 3)
        rd = torch.randn(n, 3, generator=g) * 0.3
        rd[:, 2] = -1.0
        rays = O.pack_rays(ro, rd, 2.0, 6.0, rd)
        t0 = time.perf_counter()
        with torch.no_grad():
            O.render_rays(rays, pc, pf, cfg, cfg, opt, None, chunksize=131072)
        return time.perf_counter() - t0

    one(64)
    dt = one(sample_rays)
    return dict(value=sample_rays / dt, unit="rays/s", cores=torch.get_num_threads(), kind="port", seconds=round(dt, 2),
                sample="%d rays x (64 coarse + 128 fine), 8x256 nets, forward only (torch.no_grad), after a 64-ray warm-up; "
                       "oracle/nerf_oracle.py on torch %s CPU ops" % (sample_rays, torch.__version__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(n, argv):
    """`python bench.py --gpus N` started directly (no WORLD_SIZE in the environment): become the launcher -- one rank per
    GPU under torch.distributed.run, exactly the command line the driver uses."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def kernel_kind(name):
    if "k_mlp_fwd" in name:
        return "fwd"
    if "k_mlp_dgrad" in name:
        return "dgrad"
    if "k_wgrad_bf16x3" in name:
        return "wgrad_bf16"   # (--precision bf16x3_train: the hidden x hidden blocks; "wgrad" is then the thin blocks only)
    if "k_wgrad" in name and "reduce" not in name:
        return "wgrad"
    return None


def pmc_traffic(cfg, n, kind):
    """Counter bytes per launch of kernel `kind` from the tracked rocprofv3 --pmc passes of this same command (newest
    round first; scripts/gpu_pmc.sh: separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE x2 as MI355X_MICROARCH.md
    prescribes for gfx950), or None when no pass was recorded for this configuration."""
    tag = "%dx%d_%d" % (cfg["num_layers"], cfg["hidden_size"], n)
    kname = "k_wgrad" if kind == "wgrad" else "k_mlp_%s16" % kind
    for rnd in ("r03", "r02"):
        for fn in ("%s_pmc_summary_%s.json" % (rnd, tag), "%s_pmc_summary.json" % rnd):
            path = os.path.join(ROOT, "profiles", fn)
            if not os.path.exists(path) or (fn.endswith("summary.json") and (cfg != MODEL or n != RAYS_PER_GPU)):
                continue
            rows = [v for v in json.load(open(path)).values() if v["kernel"] == kname]
            if not rows:
                continue
            # like with like: the algorithmic figure of k_wgrad is its operand READ stream, that of the forward / data-gradient
            # kernels their stash / d(pre-activation) WRITE stream (weights and masks are the small remainder)
            per = [r["fetch_gb_x2"] if kind == "wgrad" else r["write_gb"] for r in rows]
            return round(sum(per) / len(per), 3), "profiles/%s (rocprofv3 --pmc passes of this command: %s, mean over the " \
                "launch sizes recorded)" % (fn, "FETCH_SIZE x2" if kind == "wgrad" else "WRITE_SIZE")
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 20 (train) / 3 (eval)")
    ap.add_argument("--warmup", type=int, default=None, help="default 3 (train) / 1 (eval)")
    ap.add_argument("--mode", choices=("train", "eval"), default="train")
    ap.add_argument("--rays", type=int, default=RAYS_PER_GPU, help="rays per GPU and iteration (weak scaling)")
    ap.add_argument("--global-rays", type=int, default=0, help="strong scaling: rays per iteration over ALL GPUs")
    ap.add_argument("--image", type=int, default=0, help="image side: default 400 (train) / 800 (eval)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--hidden", type=int, default=MODEL["hidden_size"])
    ap.add_argument("--layers", type=int, default=MODEL["num_layers"])
    ap.add_argument("--overlap", type=int, default=-1, help="1: two-stream step (coarse backward next to the fine pass); "
                    "0: single-stream order; -1: the engine's default for the net width")
    ap.add_argument("--precision", choices=("fp32", "bf16x3", "bf16x3_fwd", "bf16x3_fwd_dgrad", "bf16x3_train"), default="fp32",
                    help="fp32 (default: the reference's arithmetic, the headline).  --mode eval --precision bf16x3: the inference "
                         "forward on the split-bf16 kernels.  --mode train --precision bf16x3_fwd: the training forward on them, "
                         "backward kernels unchanged fp32.  Both are NOT the reference's arithmetic: separate, labelled lines")
    ap.add_argument("--gather", action="store_true", help="eval: rank 0 also receives every pose's rows (output plumbing)")
    ap.add_argument("--dry-run", action="store_true", help="launcher / rendezvous / timing plumbing only, on the CPU with "
                    "gloo: no kernel runs and no number is reported (the CPU test-suite uses it)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 20 if args.mode == "train" else 3
    if args.warmup is None:
        args.warmup = 3 if args.mode == "train" else 1

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d, or "
                         "start bench.py directly and let it launch the ranks)" % (args.gpus, world, args.gpus))
    # test hook (scripts/gpu_dp2_smoke.sh): exercise the N > 1 code path on a ONE-GPU box -- every rank on cuda:0 and a
    # gloo process group (RCCL refuses two ranks on one device).  Never set by the driver.
    one_device = os.environ.get("NERFHIP_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local = 0
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.dry_run:
        return dry_run(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if not one_device and torch.cuda.device_count() < world:
        raise SystemExit("bench.py: --gpus %d but only %d device(s) visible" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if one_device:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm

    cfg = dict(MODEL, hidden_size=args.hidden, num_layers=args.layers)
    torch.manual_seed(42)  # config/lego.yml:8; every rank builds identical weights
    mc = N.FlexibleNeRFModel(**cfg).to(dev)
    mf = N.FlexibleNeRFModel(**cfg).to(dev)
    lib = N._lib.get_lib()
    side = args.image or (400 if args.mode == "train" else 800)
    focal = 0.5 * side / math.tan(0.5 * 0.6911112070083618)  # 555.5555 at 400, 1111.111 at 800 (blender camera_angle_x)
    poses = torch.stack([pose_spherical(th, -30.0, 4.0) for th in torch.linspace(-180, 180, 101)[:-1].tolist()]).to(dev)

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if (args.mode == "train" and args.precision == "bf16x3") or (args.mode == "eval" and args.precision.startswith("bf16x3_")):
        raise SystemExit("--precision bf16x3 goes with --mode eval, bf16x3_fwd / bf16x3_fwd_dgrad / bf16x3_train with --mode train")
    if args.precision.startswith("bf16x3_"):
        mc.set_training_precision(args.precision)
        mf.set_training_precision(args.precision)
    if args.mode == "train":
        strong = args.global_rays > 0
        if strong:
            lo, hi = N.parallel.shard_bounds(args.global_rays, rank, world)
            n = hi - lo
            total_rays = args.global_rays
        else:
            n = args.rays
            total_rays = n * world
        eng = N.TrainEngine(mc, mf, NC, NF, perturb=True, lindisp=False, white_background=False, noise_std=0.2, lr=5e-3,
                            seed=1234, world_size=world, rank=rank, overlap=None if args.overlap < 0 else bool(args.overlap))
        opts = N.make_options(NC, NF, num_random_rays=n)
        g = torch.Generator(device=dev).manual_seed(1000 + rank)
        images = torch.rand(8, side, side, 3, generator=g, device=dev)     # synthetic training views, resident in HBM

        def one_step(i):
            # the reference's loop body, train_nerf.py:210-270: pick a view, draw the step's distinct pixels, their rays
            # and targets (one launch, on the device), forward, loss, backward, [all-reduce], Adam with the decayed lr.
            # Weak scaling: every rank its own view; strong scaling: all ranks shard ONE view's draw.
            k = i if strong else i * world + rank
            return eng.step_on_image(images[k % 8], poses[k % poses.shape[0]], side, side, focal, opts, n,
                                     lr=N.TrainEngine.lr_at(i), global_rays=args.global_rays if strong else None)
        samples_per_step = (n * NC, n * (NC + NF))
    else:
        strong = True
        total_rays = side * side
        ex, ed = N.get_embedding_function(10, True, True), N.get_embedding_function(4, True, True)
        # eval_nerf.py:158-190 with the validation options of config/lego.yml (perturb off, noise 0); one chunk per rank
        opts = N.make_options(NC, NF, perturb=False, radiance_field_noise_std=0.0, chunksize=1 << 22)
        lo, hi = N.parallel.shard_bounds(side, rank, world)
        n = (hi - lo) * side
        mc.set_inference_precision(args.precision)
        mf.set_inference_precision(args.precision)

        def one_step(i):
            with torch.no_grad():
                out, _ = N.render_pose_rows(side, side, focal, poses[(i * 5) % poses.shape[0]], mc, mf, opts, ex, ed, rank, world)
                rgb8 = N.eval_utils._cast_to_image_device(out[3])           # eval_nerf.py:178-184 (8-bit cast on the device)
                if args.gather:
                    rgb8 = N.parallel.gather_image_rows(rgb8)
            return rgb8
        samples_per_step = (n * NC, n * (NC + NF))

    for i in range(args.warmup):
        one_step(i)
    fence()
    lib.profile_enable(1)
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        last = one_step(i)
    fence()
    dt_local = time.perf_counter() - t0
    lib.profile_enable(0)
    import ctypes
    buf = ctypes.create_string_buffer(1 << 16)
    lib.profile_report(buf, len(buf))
    clk = (ctypes.c_uint64 * 9)()
    lib.profile_clocks(clk)
    per_rank = [dt_local]
    if world > 1:
        tt = torch.tensor([dt_local], device=dev, dtype=torch.float64)
        gathered = [torch.zeros_like(tt) for _ in range(world)]
        torch.distributed.all_gather(gathered, tt)
        per_rank = [float(t) for t in gathered]
    dt = max(per_rank)
    loss_host = [float(v) for v in last.cpu()] if args.mode == "train" else None

    if rank == 0:
        kern = {}
        for line in buf.value.decode().splitlines():
            name, cnt, ms = line.rsplit(" ", 2)
            kern[name.strip("()")] = (int(cnt), float(ms))
        fwd_macs, dgrad_macs = macs_per_sample(cfg)
        m_c, m_f = samples_per_step
        # algorithmic FLOPs and HBM bytes per sample point of each MLP kernel
        flops = {"fwd": 2.0 * fwd_macs, "dgrad": 2.0 * dgrad_macs, "wgrad": 2.0 * fwd_macs}
        hbm_bytes = {"wgrad": wgrad_bytes_per_sample(cfg),
                     "fwd": stash_bytes_per_sample(cfg) if args.mode == "train" else 16 + 4,   # inference: raw out + z in
                     "dgrad": 4 * (cfg["num_layers"] * cfg["hidden_size"] + cfg["hidden_size"] + cfg["hidden_size"] // 2 + 32)
                     + 8 * (cfg["num_layers"] + 1) + 16}
        if args.precision == "bf16x3_train":
            # the weight gradient is two kernels then: the hidden x hidden blocks (k_wgrad_bf16x3: two launches per net, full and
            # half-height blocks) and the thin blocks left to the fp32 k_wgrad
            Wd, Ln = cfg["hidden_size"], cfg["num_layers"]
            big_macs = (Ln - 1) * Wd * Wd + Wd * Wd + (Wd // 2) * Wd
            big_rows = Ln * 2 * Wd + (Wd // 2 + Wd)
            flops["wgrad_bf16"], flops["wgrad"] = 2.0 * big_macs, 2.0 * (fwd_macs - big_macs)
            hbm_bytes["wgrad_bf16"], hbm_bytes["wgrad"] = 4 * big_rows, hbm_bytes["wgrad"] - 4 * big_rows
        merged = {}
        for name, (cnt, ms) in kern.items():   # (template instances of one kernel kind -- k_wgrad_bf16x3<256,256> / <128,256> -- count as one)
            kind = kernel_kind(name)
            if kind == "wgrad_bf16":
                c0, m0, n0 = merged.get(kind, (0, 0.0, name))
                merged[kind] = (c0 + cnt, m0 + ms, n0 if c0 else name)
        for kind, (cnt, ms, name) in merged.items():
            for nm in [n for n in kern if kernel_kind(n) == kind]:
                del kern[nm]
            kern["k_wgrad_bf16x3<*>"] = (cnt // 2, ms)   # (two launches = one pass over the net's blocks)
        kernels = {}
        for name, (cnt, ms) in kern.items():
            kind = kernel_kind(name)
            if kind is None:
                continue
            launches_per_step = cnt / args.steps                      # one launch per net: coarse (m_c) + fine (m_f)
            avg_ms = ms / cnt
            spl = (m_c + m_f) / launches_per_step                     # sample points per launch (mean)
            tf = flops[kind] * spl / (avg_ms * 1e-3) / 1e12
            gb = hbm_bytes[kind] * spl / 1e9
            cyc, ticks, wgs = (int(clk[3 * {"fwd": 0, "dgrad": 1, "wgrad": 2}.get(kind, 0) + c]) if kind in ("fwd", "dgrad", "wgrad") else 0
                               for c in range(3))
            ghz = 0.1 * cyc / ticks if ticks else None
            counter_gb, source = pmc_traffic(cfg, n, kind) if (args.mode == "train" and args.precision == "fp32") else (None, None)
            # a kernel is priced against ITS OWN matrix pipe: fp32 MFMA, or the bf16 MFMA at three instructions per product block
            peak = BF16X3_PEAK_TFLOPS if "bf16x3" in name else FP32_MFMA_PEAK_TFLOPS
            kernels[kind] = dict(kernel=name, ms_per_step=round(ms / args.steps, 4), avg_launch_ms=round(avg_ms, 4), launches=cnt,
                                 algorithmic_gflop_per_launch=round(flops[kind] * spl / 1e9, 2), tflops=round(tf, 2),
                                 peak=round(peak, 1), frac=round(tf / peak, 4),
                                 sclk_ghz=None if ghz is None else round(ghz, 3),
                                 frac_at_measured_clock=None if ghz is None else round(tf / (FP32_MFMA_PEAK_TFLOPS * ghz / PEAK_CLOCK_GHZ), 4),
                                 traffic=dict(algorithmic_gb=round(gb, 3), counter_gb=counter_gb, source=source),
                                 hbm_tb_s=round(gb / avg_ms, 3), hbm_frac=round(gb / avg_ms / HBM_PEAK_TBS, 4))
        roof = None
        if kernels:
            dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"])   # the kernel the step spends most time in
            lowest = min(kernels, key=lambda k: kernels[k]["frac"])
            d = kernels[dom]
            roof = dict(bound="mfma", kernel=d["kernel"], achieved=d["tflops"], peak=d["peak"], unit="TFLOP/s",
                        frac=d["frac"], traffic=d["traffic"], avg_launch_ms=d["avg_launch_ms"], launches=d["launches"],
                        algorithmic_gflop_per_launch=d["algorithmic_gflop_per_launch"],
                        sclk_ghz=d["sclk_ghz"], frac_at_measured_clock=d["frac_at_measured_clock"],
                        hbm=dict(achieved_tb_s=d["hbm_tb_s"], peak_tb_s=HBM_PEAK_TBS, frac=d["hbm_frac"]),
                        lowest_frac_kernel=kernels[lowest]["kernel"], lowest_frac=kernels[lowest]["frac"],
                        mlp_kernels=kernels,
                        kernel_ms_per_step={nm: round(m / args.steps, 4) for nm, (_, m) in sorted(kern.items(), key=lambda kv: -kv[1][1])})
        if args.mode == "train":
            total_flops = (2.0 * (2 * fwd_macs + dgrad_macs)) * (m_c + m_f)
            step_bytes = sum(hbm_bytes[k] for k in ("fwd", "dgrad", "wgrad")) * (m_c + m_f)
            workload = ("lego %dx%d synthetic views (BASELINE configs[%d]): %d rays/GPU/iter (%d over all GPUs), %d coarse + %d "
                        "fine samples, %dx%d coarse+fine nets, perturb, noise 0.2, Adam, full iteration"
                        % (side, side, 2 if (strong and side == 800) else 1, n, total_rays, NC, NF, cfg["num_layers"], cfg["hidden_size"]))
            metric = "train rays/sec"
        else:
            total_flops = 2.0 * fwd_macs * (m_c + m_f)
            step_bytes = hbm_bytes["fwd"] * (m_c + m_f)
            workload = ("eval_nerf.py 360-degree render (BASELINE configs[4]): %dx%d poses, rows sharded over %d GPU(s) (%d rays/GPU/"
                        "pose), %d coarse + %d fine samples, %dx%d nets, perturb off, 8-bit cast on the device, one step = one pose"
                        % (side, side, world, n, NC, NF, cfg["num_layers"], cfg["hidden_size"]))
            metric = "eval rays/sec"
        sec = dt / args.steps
        res = dict(metric=metric, value=round(total_rays * args.steps / dt, 2), unit="rays/s", n_gpus=world,
                   steps=args.steps, warmup=args.warmup, ms_per_step=round(sec * 1e3, 3),
                   ms_per_step_per_rank=[round(t / args.steps * 1e3, 3) for t in per_rank],
                   higher_is_better=True, scaling="strong" if strong else "weak", vs_baseline=None,
                   dtype="f32" if args.precision == "fp32" else
                   ("bf16x3 (fp32 operands split into two bf16 pieces, three bf16 MFMAs per product block, f32 accumulate; "
                    "fp32-equivalent FLOPs)" if args.precision == "bf16x3" else
                    ("forward bf16x3 (split-bf16 products, f32 accumulate), backward + optimizer f32; fp32-equivalent FLOPs"
                     if args.precision == "bf16x3_fwd" else
                     ("forward + data gradient bf16x3 (split-bf16 products, f32 accumulate), weight gradient + optimizer f32; "
                      "fp32-equivalent FLOPs" if args.precision == "bf16x3_fwd_dgrad" else
                      "forward, data gradient and the hidden x hidden weight-gradient blocks bf16x3 (split-bf16 products, f32 "
                      "accumulate); thin weight-gradient blocks + optimizer f32; fp32-equivalent FLOPs"))),
                   data="synthetic",
                   config=dict(workload=workload, rays_per_gpu=n, global_rays=total_rays, parallelism="dp%d" % world,
                               two_stream_step=bool(eng.overlap) if args.mode == "train" else None,
                               backend=("gloo(one-device test hook)" if one_device else "nccl(RCCL)") if world > 1 else None),
                   step_tflops=round(total_flops / sec / 1e12, 2),
                   step_frac_of_fp32_mfma_peak=round(total_flops / sec / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                   step_algorithmic_hbm_tb_s=round(step_bytes / sec / 1e12, 3),
                   step_hbm_frac_of_8tb_s=round(step_bytes / sec / 1e12 / HBM_PEAK_TBS, 4),
                   final_loss=loss_host, roofline=roof)
        if world == 1 and not args.no_cpu_baseline:
            if args.mode == "train":
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
                res["cpu_baseline"] = cpu_baseline_eval()
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res))
    if world > 1:
        torch.distributed.destroy_process_group()


def dry_run(args, world, rank):
    """The plumbing around the timed region without a GPU: rendezvous (gloo), barrier, per-rank timing gathered to rank 0,
    ONE JSON line with value null.  Exists so that the CPU test-suite can start `python bench.py --gpus 2 --dry-run`
    exactly the way the driver starts the real thing."""
    if world > 1:
        torch.distributed.init_process_group("gloo")
        torch.distributed.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    per_rank = [time.perf_counter() - t0]
    if world > 1:
        tt = torch.tensor([per_rank[0]], dtype=torch.float64)
        gathered = [torch.zeros_like(tt) for _ in range(world)]
        torch.distributed.all_gather(gathered, tt)
        per_rank = [float(t) for t in gathered]
        lo, hi = N.parallel.shard_bounds(args.global_rays or args.rays * world, rank, world)
        cover = torch.tensor([hi - lo], dtype=torch.int64)
        torch.distributed.all_reduce(cover)
        assert int(cover) == (args.global_rays or args.rays * world)
    if rank == 0:
        print(json.dumps(dict(metric="train rays/sec" if args.mode == "train" else "eval rays/sec", value=None, unit="rays/s",
                              n_gpus=world, steps=args.steps, warmup=args.warmup, dry_run=True,
                              ms_per_step_per_rank=[round(t * 1e3, 3) for t in per_rank],
                              scaling="strong" if (args.global_rays or args.mode == "eval") else "weak")))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
