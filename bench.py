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
F16X3_PEAK_TFLOPS = 2500.0 / 3.0   # dense fp16 MFMA peak (same table) / three MFMAs per fp32-equivalent product block
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


def _synthetic_batch(O, wl, cfg, n, g):
    """n rays of the workload's geometry + targets + the reference's four random draws (CPU tensors)."""
    nc, nf = wl["nc"], wl["nf"]
    if wl["no_ndc"]:
        ro = torch.tensor([0.0, 0.0, 4.0]).expand(n, 3)
        rd = torch.randn(n, 3, generator=g) * 0.3
        rd[:, 2] = -1.0
        rays = O.pack_rays(ro, rd, wl["near"], wl["far"], rd)
    else:  # forward-facing capture: NDC rays (nerf/train_utils.py:156-160), viewdirs from the pre-NDC directions
        ro = torch.tensor([0.0, 0.0, 0.3]).expand(n, 3) + 0.05 * torch.randn(n, 3, generator=g)
        rd = torch.randn(n, 3, generator=g) * 0.3
        rd[:, 2] = -1.0
        no, nd = O.ndc_rays(wl["H"], wl["W"], wl["focal"], 1.0, ro, rd)
        rays = O.pack_rays(no, nd, wl["near"], wl["far"], rd)
    tgt = torch.rand(n, 3, generator=g)
    rand = dict(t_rand=torch.rand(n, nc, generator=g), noise_coarse=torch.randn(n, nc, generator=g),
                u=torch.rand(n, nf, generator=g), noise_fine=torch.randn(n, nc + nf, generator=g))
    opt = dict(num_coarse=nc, num_fine=nf, perturb=True, lindisp=False, white_background=False, noise_std=wl["noise"])
    return rays, tgt, rand, opt


def cpu_baseline(sample_rays=RAYS_PER_GPU, wl=None, cfg=None):
    """The oracle (kind "port": oracle/nerf_oracle.py, the CPU restatement of the reference path, bit-identical to the
    reference's own functions on CPU -- tests/test_oracle.py) forward + backward on ONE full batch of `sample_rays`
    synthetic rays of the workload, after a 64-ray warm-up (thread pool, allocator)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nerf_oracle as O
    wl = wl or WORKLOADS["lego"]
    cfg = dict(cfg or wl["model"])
    # 16-32 threads is the fastest setting on the 256-thread EPYC host of the GPU box (measured: 8 -> 520, 16 -> 709,
    # 32 -> 576, 64 -> 253, 128 -> 49 rays/s at 128 rays; profiles/r01_cpu_threads.txt): more threads only add
    # fork/join overhead
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    pc = {k: v.requires_grad_(True) for k, v in O.init_params(cfg, seed=1).items()}
    pf = {k: v.requires_grad_(True) for k, v in O.init_params(cfg, seed=2).items()}
    g = torch.Generator().manual_seed(0)

    def one(n):
        rays, tgt, rand, opt = _synthetic_batch(O, wl, cfg, n, g)
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
                sample="one full batch of %d rays x (%d coarse + %d fine), %dx%d nets, fwd+bwd (no optimizer), after a 64-ray "
                       "warm-up; oracle/nerf_oracle.py = the reference's functions restated on torch %s CPU ops (bit-identical "
                       "to the reference on CPU, tests/test_oracle.py)" % (n, wl["nc"], wl["nf"], cfg["num_layers"], cfg["hidden_size"],
                                                                           torch.__version__))


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


def pytorch_rocm_reference(dev, n=RAYS_PER_GPU, reps=3, wl=None, cfg=None):
    """The reference's own PyTorch path (the oracle's torch ops, op for op) on THIS GPU: forward+backward on n rays --
    the denominator of the north star's "x the reference single-GPU PyTorch-ROCm rays/sec"."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nerf_oracle as O
    wl = wl or WORKLOADS["lego"]
    cfg = dict(cfg or wl["model"])
    nc, nf = wl["nc"], wl["nf"]
    pc = {k: v.to(dev).requires_grad_(True) for k, v in O.init_params(cfg, 1).items()}
    pf = {k: v.to(dev).requires_grad_(True) for k, v in O.init_params(cfg, 2).items()}
    g = torch.Generator().manual_seed(0)
    rays, tgt, _, opt = _synthetic_batch(O, wl, cfg, n, g)
    rays, tgt = rays.to(dev), tgt.to(dev)
    best = float("inf")
    for it in range(reps + 1):
        rand = dict(t_rand=torch.rand(n, nc, device=dev), noise_coarse=torch.randn(n, nc, device=dev),
                    u=torch.rand(n, nf, device=dev), noise_fine=torch.randn(n, nc + nf, device=dev))
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
    if "k_bwd64r_reduce" in name:
        return None
    if "k_bwd64r" in name:
        return "bwd64r"       # (--compact fused / fused_compact / fused_stash: [forward recomputed +] data gradient + weight gradient, one kernel)
    if "k_mlp_fwd" in name or "k_fwd64r" in name:
        return "fwd"
    if "k_mlp_dgrad" in name:
        return "dgrad"
    if "k_wgrad_f16x3" in name:
        return "wgrad_big"    # (--precision f16x3_train: the hidden x hidden blocks on the fp16 MFMAs)
    if "k_wgrad<" in name and "[thin" in name:
        return "wgrad_thin"   # (... and what is left to the fp32 kernel then)
    if "k_wgrad<" in name:
        return "wgrad"
    return None


def kernel_family(name):
    """The matrix pipe a kernel multiplies on: "fp32" (v_mfma_f32_16x16x4 / 32x32x2) or the fp16 MFMAs on split operands."""
    if "f16x3" in name:
        return "f16x3"
    return "fp32"


def precision_level(prec):
    """(piece format | None, level) of a --precision value: level 0 fp32, 1 inference forward, 2 + training forward, 3 + data
    gradient, 4 + the large weight-gradient blocks (include/nerfhip.h NERFHIP_PRECISION_*)."""
    if prec == "fp32":
        return None, 0
    fmt, _, rest = prec.partition("x3")
    return fmt + "x3", {"": 1, "_fwd": 2, "_fwd_dgrad": 3, "_train": 4}[rest]


def lib_sources_sha16():
    """Fingerprint of the kernel sources the library was built from (what a tracked PMC summary must have been measured on)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "nerf-pytorch_amd", "csrc")
    for p in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h")) + glob.glob(os.path.join(d, "*.cpp")) +
                    [os.path.join(d, "Makefile"), os.path.join(ROOT, "include", "nerfhip.h")]):
        h.update(os.path.basename(p).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(cfg, n, kind):
    """Counter bytes per launch of kernel `kind` from the tracked rocprofv3 --pmc passes of this same command (newest
    round first; scripts/gpu_pmc.sh: separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE x2 as MI355X_MICROARCH.md
    prescribes for gfx950), or None when no pass was recorded for this configuration.  A summary is used only if it carries
    the fingerprint of the kernel sources this library was built from (`lib_sources_sha16`): a stale file is refused and the
    line says so."""
    tag = "%dx%d_%d" % (cfg["num_layers"], cfg["hidden_size"], n)
    # (r64: the 64-wide nets' stashed fused backward, csrc/mlp64r.hip mode 5 -- k_fwd64r writes the register-image stash, k_bwd64r reads it)
    r64 = kind in ("fwd64r", "bwd64r")
    kname = "k_wgrad" if kind == "wgrad" else ("k_" + kind if r64 else "k_mlp_%s16" % kind)
    mine = lib_sources_sha16()
    stale = None
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        for fn in ("%s_pmc_summary_%s.json" % (rnd, tag), "%s_pmc_summary.json" % rnd):
            path = os.path.join(ROOT, "profiles", fn)
            if not os.path.exists(path) or (fn.endswith("summary.json") and (cfg != MODEL or n != RAYS_PER_GPU)):
                continue
            doc = json.load(open(path))
            stamp = doc.get("_lib_sources_sha16") if isinstance(doc.get("_lib_sources_sha16"), str) else None
            if stamp != mine:
                stale = stale or "profiles/%s refused: measured on kernel sources %s, this library is %s" % (fn, stamp or "(unstamped)", mine)
                continue
            rows = [v for v in doc.values() if isinstance(v, dict) and v.get("kernel") == kname]
            if not rows:
                continue
            # like with like: the algorithmic figure of k_wgrad is its operand READ stream, that of the forward / data-gradient
            # kernels their stash / d(pre-activation) WRITE stream (weights and masks are the small remainder)
            reads = kind in ("wgrad", "bwd64r")
            per = [r["fetch_gb_x2"] if reads else r["write_gb"] for r in rows]
            return round(sum(per) / len(per), 3), "profiles/%s (rocprofv3 --pmc passes of this command on kernel sources %s: %s, mean over " \
                "the launch sizes recorded)" % (fn, mine, "FETCH_SIZE x2" if reads else "WRITE_SIZE")
    return None, stale


def fern_poses(k=40):
    """Forward-facing camera-to-world matrices like an LLFF capture (load_llff.py poses after recentering): the camera
    looks down -z, a few degrees of rotation and a few tenths of a unit of translation around the origin."""
    out = []
    for i in range(k):
        a, b = 0.12 * math.sin(2 * math.pi * i / k), 0.08 * math.cos(2 * math.pi * i / k)
        ry = torch.tensor([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
        rx = torch.tensor([[1, 0, 0], [0, math.cos(b), -math.sin(b)], [0, math.sin(b), math.cos(b)]])
        c2w = torch.eye(4)
        c2w[:3, :3] = ry @ rx
        c2w[:3, 3] = torch.tensor([0.3 * math.sin(2 * math.pi * i / k), 0.2 * math.cos(2 * math.pi * i / k), 0.1 * math.sin(4 * math.pi * i / k)])
        out.append(c2w)
    return torch.stack(out)


# BASELINE.json configs[1] (config/lego.yml: the configuration the metric is quoted on) and configs[3] (config/fern.yml:
# models 4x64 skip 3 with 6 xyz frequencies :46-58, NDC rays with near 0 / far 1 :11-14, sigma noise 1.0 :89; 64 + 64 samples as
# BASELINE.json states the config -- the yml itself says num_fine 128; LLFF fern at downsample 8: 378 x 504, focal 407.5)
WORKLOADS = {
    "lego": dict(H=400, W=400, focal=FOCAL, nc=64, nf=128, model=MODEL, near=2.0, far=6.0, no_ndc=True, noise=0.2, baseline_config=1),
    "fern": dict(H=378, W=504, focal=407.5, nc=64, nf=64,
                 model=dict(num_layers=4, hidden_size=64, skip_connect_every=3, num_encoding_fn_xyz=6, num_encoding_fn_dir=4),
                 near=0.0, far=1.0, no_ndc=False, noise=1.0, baseline_config=3),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 20 (train) / 3 (eval)")
    ap.add_argument("--warmup", type=int, default=None, help="default 3 (train) / 1 (eval)")
    ap.add_argument("--mode", choices=("train", "eval"), default="train")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="lego",
                    help="lego: BASELINE configs[1] (the headline; default).  fern: BASELINE configs[3] as config/fern.yml declares it "
                         "(4x64 nets, skip 3, 6 xyz frequencies, NDC rays, near 0 / far 1, 378x504, sigma noise 1.0, 64 + 64 samples)")
    ap.add_argument("--rays", type=int, default=RAYS_PER_GPU, help="rays per GPU and iteration (weak scaling)")
    ap.add_argument("--global-rays", type=int, default=0, help="strong scaling: rays per iteration over ALL GPUs")
    ap.add_argument("--image", type=int, default=0, help="image side: default 400 (train) / 800 (eval)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-labelled-lines", action="store_true", help="default run only: skip the child-process lines (other precisions / workloads)")
    ap.add_argument("--hidden", type=int, default=0, help="hidden_size (default: the workload's)")
    ap.add_argument("--layers", type=int, default=0, help="num_layers (default: the workload's)")
    ap.add_argument("--overlap", type=int, default=-1, help="1: two-stream step (coarse backward next to the fine pass); "
                    "0: single-stream order; -1: the engine's default for the net width")
    ap.add_argument("--precision", choices=("fp32", "f16x3", "f16x3_fwd", "f16x3_fwd_dgrad", "f16x3_train", "fp32+f16x3_train"),
                    default="fp32",
                    help="fp32 (default: the reference's arithmetic, the headline).  f16x3*: the GEMMs on the fp16 MFMAs with every "
                         "operand split into two IEEE fp16 pieces (fp32-grade products): --mode eval --precision f16x3 the inference "
                         "forward; --mode train --precision f16x3_fwd the training forward, f16x3_fwd_dgrad + the data-gradient chain, "
                         "f16x3_train + the large weight-gradient blocks.  A+B: coarse net A, fine net B.  Separate, labelled lines: "
                         "the driver's default stays fp32")
    ap.add_argument("--compact", nargs="?", const="gather", default=None, choices=("dense", "gather", "recompute", "fused", "fused_compact", "fused_stash", "auto"),
                    help="train: compacted backward (FlexibleNeRFModel.set_backward_compaction): data and weight gradient over the sample "
                         "points whose d(loss)/d(raw) row is not all zero; `recompute`: additionally a stash-free training forward, the "
                         "backward re-runs the forward for the kept samples; `auto`: TrainEngine(backward='auto') picks dense / compacted / recomputed per net "
                         "and step from the zero fraction the previous steps reported.  A labelled line: it states the zero fraction of the step it "
                         "timed and prices the backward kernels on the FLOPs they executed; the driver's default stays dense")
    ap.add_argument("--gather", action="store_true", help="eval: rank 0 also receives every pose's rows (output plumbing)")
    ap.add_argument("--no-kernel-profile", action="store_true", help="do not bracket the launches of the timed region with HIP events (no "
                    "per-kernel times, no roofline object): what the events cost a short step")
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
    rccl = None
    if world > 1:
        if one_device:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm
            try:
                rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                rccl = "unknown"

    wl = dict(WORKLOADS[args.workload])
    cfg = dict(wl["model"])
    if args.hidden:
        cfg["hidden_size"] = args.hidden
    if args.layers:
        cfg["num_layers"] = args.layers
    nc, nf = wl["nc"], wl["nf"]
    dx, dd = 3 + 6 * cfg["num_encoding_fn_xyz"], 3 + 6 * cfg["num_encoding_fn_dir"]
    torch.manual_seed(42)  # config/lego.yml:8; every rank builds identical weights
    mc = N.FlexibleNeRFModel(**cfg).to(dev)
    mf = N.FlexibleNeRFModel(**cfg).to(dev)
    lib = N._lib.get_lib()
    if args.workload == "lego":
        side = args.image or (400 if args.mode == "train" else 800)
        height = width = side
        focal = 0.5 * side / math.tan(0.5 * 0.6911112070083618)  # 555.5555 at 400, 1111.111 at 800 (blender camera_angle_x)
        poses = torch.stack([pose_spherical(th, -30.0, 4.0) for th in torch.linspace(-180, 180, 101)[:-1].tolist()]).to(dev)
    else:
        height, width, focal = wl["H"], wl["W"], wl["focal"]
        side = 0
        poses = fern_poses().to(dev)

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    prec_c, prec_f = args.precision.split("+") if "+" in args.precision else (args.precision, args.precision)
    infer_only = prec_f == "f16x3"
    if (args.mode == "train" and infer_only) or (args.mode == "eval" and not infer_only and prec_f != "fp32"):
        raise SystemExit("--precision f16x3 goes with --mode eval, f16x3_fwd / _fwd_dgrad / _train with --mode train")
    if (prec_c != "fp32" or prec_f != "fp32") and cfg["hidden_size"] > 256:
        # (nerfhip_plan_create_ex refuses such plans: the fp16-piece kernels exist for the 64-, 128- and 256-wide kernel widths)
        raise SystemExit("--precision %s needs hidden_size <= 256 (this workload: %d); the 512-wide nets run fp32"
                         % (args.precision, cfg["hidden_size"]))
    if args.mode == "train":
        if prec_c != "fp32":
            mc.set_training_precision(prec_c)
        if prec_f != "fp32":
            mf.set_training_precision(prec_f)
        if args.compact in ("dense", "gather", "recompute", "fused", "fused_compact", "fused_stash"):
            # (fused / fused_compact: the one-kernel backward of 64-wide fp32 nets, csrc/mlp64r.hip -- raises for other geometries; it
            # is those nets' default: `dense` asks for the three-kernel backward)
            mc.set_backward_compaction({"dense": False, "gather": True}.get(args.compact, args.compact))
            mf.set_backward_compaction({"dense": False, "gather": True}.get(args.compact, args.compact))
        strong = args.global_rays > 0
        if strong:
            lo, hi = N.parallel.shard_bounds(args.global_rays, rank, world)
            n = hi - lo
            total_rays = args.global_rays
        else:
            n = args.rays
            total_rays = n * world
        eng = N.TrainEngine(mc, mf, nc, nf, perturb=True, lindisp=False, white_background=False, noise_std=wl["noise"], lr=5e-3,
                            seed=1234, world_size=world, rank=rank, overlap=None if args.overlap < 0 else bool(args.overlap),
                            backward="auto" if args.compact == "auto" else None)
        opts = N.make_options(nc, nf, num_random_rays=n, radiance_field_noise_std=wl["noise"], no_ndc=wl["no_ndc"], near=wl["near"],
                              far=wl["far"])
        g = torch.Generator(device=dev).manual_seed(1000 + rank)
        images = torch.rand(8, height, width, 3, generator=g, device=dev)     # synthetic training views, resident in HBM

        def one_step(i):
            # the reference's loop body, train_nerf.py:210-270: pick a view, draw the step's distinct pixels, their rays
            # and targets (one launch, on the device), forward, loss, backward, [all-reduce], Adam with the decayed lr.
            # Weak scaling: every rank its own view; strong scaling: all ranks shard ONE view's draw.
            k = i if strong else i * world + rank
            return eng.step_on_image(images[k % 8], poses[k % poses.shape[0]], height, width, focal, opts, n,
                                     lr=N.TrainEngine.lr_at(i), global_rays=args.global_rays if strong else None)
        samples_per_step = (n * nc, n * (nc + nf))
    else:
        if args.workload != "lego":
            raise SystemExit("--mode eval is BASELINE configs[4] (lego 800x800)")
        strong = True
        total_rays = side * side
        ex, ed = N.get_embedding_function(10, True, True), N.get_embedding_function(4, True, True)
        # eval_nerf.py:158-190 with the validation options of config/lego.yml (perturb off, noise 0); one chunk per rank
        opts = N.make_options(nc, nf, perturb=False, radiance_field_noise_std=0.0, chunksize=1 << 22)
        lo, hi = N.parallel.shard_bounds(side, rank, world)
        n = (hi - lo) * side
        mc.set_inference_precision(prec_c)
        mf.set_inference_precision(prec_f)

        def one_step(i):
            with torch.no_grad():
                out, _ = N.render_pose_rows(side, side, focal, poses[(i * 5) % poses.shape[0]], mc, mf, opts, ex, ed, rank, world)
                rgb8 = N.eval_utils._cast_to_image_device(out[3])           # eval_nerf.py:178-184 (8-bit cast on the device)
                if args.gather:
                    rgb8 = N.parallel.gather_image_rows(rgb8)
            return rgb8
        samples_per_step = (n * nc, n * (nc + nf))

    for i in range(args.warmup):
        one_step(i)
    if not args.no_kernel_profile:
        lib.profile_reserve(96 * args.steps)          # a step is 40-70 launches; event creation stays out of the timed region
    fence()
    lib.profile_enable(0 if args.no_kernel_profile else 1)
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        last = one_step(i)
    dt_issue = time.perf_counter() - t0                # the host is done enqueueing; == dt_local when the step is host-bound
    fence()
    dt_local = time.perf_counter() - t0
    lib.profile_enable(0)
    import ctypes
    buf = ctypes.create_string_buffer(1 << 16)
    lib.profile_report(buf, len(buf))
    clk = (ctypes.c_uint64 * 9)()
    lib.profile_clocks(clk)
    per_rank = [dt_local]
    allreduce_ms = None
    if world > 1:
        tt = torch.tensor([dt_local], device=dev, dtype=torch.float64)
        gathered = [torch.zeros_like(tt) for _ in range(world)]
        torch.distributed.all_gather(gathered, tt)
        per_rank = [float(t) for t in gathered]
        if args.mode == "train":
            allreduce_ms = eng.collective_times_ms()
    dt = max(per_rank)
    loss_host = [float(v) for v in last.cpu()] if args.mode == "train" else None
    # compacted backward: the sample points the LAST timed step's backward kept, per net (two words per net of its workspace)
    kept = eng.backward_sample_counts() if (args.mode == "train" and args.compact not in (None, "dense")) else None
    # (the mode the accounting below assumes: `auto` -> what the last timed step's nets ran in; per-net / per-step mixes make the auto
    # line's per-kernel fractions approximate -- its rays/s is what it is)
    eff_compact = None if args.compact == "dense" else args.compact
    if args.mode == "train" and args.compact in (None, "auto"):   # (None: the models' defaults -- fused where a plan has it)
        modes = [m.backward_compaction for m in (mc, mf)]
        eff_compact = ("fused_compact" if 4 in modes else "fused_stash" if 5 in modes else "fused" if 3 in modes else "recompute" if 2 in modes
                       else ("gather" if 1 in modes else None))
    # What the per-launch HIP events of the timed region cost: the same K steps once more WITHOUT them (N = 1 only).  Nothing at
    # 27 ms per step; 0.28 ms of a 2.0-ms fern step (a step is ~40 launches, each with two event records on the host's path).
    unprofiled = None
    if world == 1 and not args.no_kernel_profile:
        fence()
        t1 = time.perf_counter()
        for i in range(args.warmup + args.steps, args.warmup + 2 * args.steps):
            one_step(i)
        dt_plain_issue = time.perf_counter() - t1
        fence()
        dt_plain = time.perf_counter() - t1
        unprofiled = dict(value=round(total_rays * args.steps / dt_plain, 2), unit="rays/s", ms_per_step=round(dt_plain / args.steps * 1e3, 3),
                          host_issue_ms_per_step=round(dt_plain_issue / args.steps * 1e3, 3),
                          what="the same %d steps right after the timed region, without the per-launch HIP events (no per-kernel times)" % args.steps)

    if rank == 0:
        kern = {}
        for line in buf.value.decode().splitlines():
            name, cnt, ms = line.rsplit(" ", 2)
            kern[name.strip("()")] = (int(cnt), float(ms))
        fwd_macs, dgrad_macs = macs_per_sample(cfg, dx, dd)
        m_c, m_f = samples_per_step
        Wd, Ln = cfg["hidden_size"], cfg["num_layers"]
        # algorithmic FLOPs and HBM bytes per sample point of each MLP kernel
        stash_b = stash_bytes_per_sample(cfg) if args.mode == "train" else 16 + 4   # inference: raw out + z in
        dgrad_b = 4 * (Ln * Wd + Wd + Wd // 2 + 32) + 8 * (Ln + 1) + 16
        wgrad_b = wgrad_bytes_per_sample(cfg)
        # the hidden x hidden blocks of the level-4 plans: layers_xyz, fc_feat and -- 256-wide kernels only -- the hidden columns of layers_dir
        # ... and the thin blocks that ride on them as guests (wgrad_f16.hip SA / SB): a skip layer's xyz columns, fc_alpha's row,
        # the direction columns -- their own regions are the only bytes they add
        wide = Wd > 128
        n_skip = sum(1 for i in range(1, Ln - 1) if i % cfg["skip_connect_every"] == 0)
        big_macs = (Ln - 1) * Wd * Wd + Wd * Wd + ((Wd // 2) * Wd if wide else 0) + n_skip * Wd * dx + Wd + ((Wd // 2) * dd if wide else 0)
        big_b = 4 * (Ln * 2 * Wd + ((Wd // 2 + Wd) if wide else 0) + n_skip * 64 + 32 + (32 if wide else 0))
        # what is left to the fp32 kernel then: layer1's block, fc_rgb's, and -- 128-wide kernels -- layers_dir's two
        thin_b = 4 * ((Wd + 64) + (32 + Wd // 2) + (0 if wide else (Wd // 2 + Wd) + (Wd // 2 + 32)))
        big_launches = 1 + (1 if n_skip else 0) + 1 + (1 if wide else 0)   # (one launch per block shape: plain, + xyz guest, + fc_alpha guest, half-height)
        # which kernel each net's passes run on: (kind, family) -> [flops per step, bytes per step, launches per step]
        work = {}

        def add(kind, fam, flops, nbytes, launches=1):
            w = work.setdefault((kind, fam), [0.0, 0.0, 0])
            w[0] += flops
            w[1] += nbytes
            w[2] += launches
        # compacted backward (--compact): the backward kernels are priced on the sample points they EXECUTED -- those the last timed
        # step's backward kept -- so that a fraction of a roofline can never exceed 1 by skipping work
        kept_c, kept_f = m_c, m_f
        if kept is not None:
            kept_c = kept["coarse"][0] if kept["coarse"] else m_c
            kept_f = kept["fine"][0] if kept["fine"] else m_f
        for m, mb, prec in ((m_c, kept_c, prec_c), (m_f, kept_f, prec_f)):
            fmt, level = precision_level(prec)
            r64_stash_b = 4 * (64 * Ln + 192)   # (the register-image stash of mode 5, csrc/nh_r64.h: bytes per sample point)
            if args.mode == "train" and eff_compact in ("fused", "fused_compact"):
                add("fwd", "fp32", 2.0 * fwd_macs * m, 20 * m)   # (stash-free)
            elif args.mode == "train" and eff_compact == "fused_stash":
                add("fwd", "fp32", 2.0 * fwd_macs * m, (20 + r64_stash_b) * m)
            elif args.mode == "train" and eff_compact == "recompute":  # (stash-free pass over all samples + a stash-writing pass over the kept ones)
                add("fwd", fmt if level >= 1 else "fp32", 2.0 * fwd_macs * (m + mb), 20 * m + stash_b * mb, 2)
            else:
                add("fwd", fmt if level >= 1 else "fp32", 2.0 * fwd_macs * m, stash_b * m)
            if args.mode != "train":
                continue
            if eff_compact in ("fused", "fused_compact"):
                # the stash-free forward above; then ONE kernel: the forward again, the data gradient, the weight gradient -- executed
                # FLOPs; it reads a sample's depth, ray and d(raw) row and writes one partial per workgroup
                # (the recomputed forward stops at the activations: fc_alpha's and fc_rgb's rows are not multiplied again)
                add("bwd64r", "fp32", 2.0 * (2 * fwd_macs + dgrad_macs - (Wd + 3 * (Wd // 2))) * mb, 40 * mb)
                continue
            if eff_compact == "fused_stash":
                # ONE kernel: the data gradient and the weight gradient over the register-image stash the forward left (nothing recomputed)
                add("bwd64r", "fp32", 2.0 * (fwd_macs + dgrad_macs) * mb, (16 + r64_stash_b) * mb)
                continue
            add("dgrad", fmt if level >= 3 else "fp32", 2.0 * dgrad_macs * mb, dgrad_b * mb)
            if level == 4 and 64 < Wd <= 256:
                add("wgrad_big", fmt, 2.0 * big_macs * mb, big_b * mb, big_launches)
                add("wgrad_thin", "fp32", 2.0 * (fwd_macs - big_macs) * mb, thin_b * mb)
            else:
                add("wgrad", "fp32", 2.0 * fwd_macs * mb, wgrad_b * mb)
        # template instances of one kernel (k_wgrad_f16x3<full> / <half>) count as one
        merged = {}
        for name, (cnt, ms) in kern.items():
            kind = kernel_kind(name)
            if kind is None:
                continue
            key = (kind, kernel_family(name))
            c0, m0, names = merged.get(key, (0, 0.0, []))
            merged[key] = (c0 + cnt, m0 + ms, names + [name])
        kernels = {}
        kinds_seen = [k for k, _ in merged]
        for (kind, fam), (cnt, ms, names) in merged.items():
            if (kind, fam) not in work:
                continue
            fl, by, launches = work[(kind, fam)]
            label = names[0] if len(names) == 1 else names[0].split("<")[0] + "<*>"
            ms_step = ms / args.steps
            tf = fl / (ms_step * 1e-3) / 1e12
            gb = by / 1e9
            ghz = None
            if fam == "fp32" and kind in ("fwd", "dgrad", "wgrad", "wgrad_thin", "bwd64r"):
                cyc, ticks = (int(clk[3 * {"fwd": 0, "dgrad": 1, "bwd64r": 1}.get(kind, 2) + c]) for c in range(2))
                ghz = 0.1 * cyc / ticks if ticks else None
            counter_gb, source = (pmc_traffic(cfg, n, kind) if (args.mode == "train" and args.precision == "fp32" and args.workload == "lego"
                                                               and kind in ("fwd", "dgrad", "wgrad")) else (None, None))
            if args.mode == "train" and args.workload == "fern" and eff_compact == "fused_stash" and kind in ("fwd", "bwd64r"):
                counter_gb, source = pmc_traffic(cfg, n, "fwd64r" if kind == "fwd" else kind)
            # a kernel is priced against ITS OWN matrix pipe: fp32 MFMA, or the 16-bit MFMA at three instructions per product block
            peak = FP32_MFMA_PEAK_TFLOPS if fam == "fp32" else F16X3_PEAK_TFLOPS
            key = kind if kinds_seen.count(kind) == 1 else "%s[%s]" % (kind, fam)
            kernels[key] = dict(kernel=label, family=fam, ms_per_step=round(ms_step, 4), avg_launch_ms=round(ms / cnt, 4), launches=cnt,
                                launches_per_step=launches, algorithmic_gflop_per_step=round(fl / 1e9, 2),
                                algorithmic_gflop_per_launch=round(fl / launches / 1e9, 2), tflops=round(tf, 2),
                                peak=round(peak, 1), frac=round(tf / peak, 4),
                                sclk_ghz=None if ghz is None else round(ghz, 3),
                                frac_at_measured_clock=None if ghz is None else round(tf / (FP32_MFMA_PEAK_TFLOPS * ghz / PEAK_CLOCK_GHZ), 4),
                                traffic=dict(algorithmic_gb=round(gb / launches, 3), counter_gb=counter_gb, source=source),
                                hbm_tb_s=round(gb / ms_step, 3), hbm_frac=round(gb / ms_step / HBM_PEAK_TBS, 4))
        roof = None
        if kernels:
            dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"])   # the kernel the step spends most time in
            lowest = min(kernels, key=lambda k: kernels[k]["frac"])
            d = kernels[dom]
            # the narrow nets are within reach of both roofs: name the one the dominant kernel is closer to
            bound = "hbm" if d["hbm_frac"] > d["frac"] else "mfma"
            roof = dict(bound=bound, kernel=d["kernel"],
                        achieved=d["tflops"] if bound == "mfma" else round(d["hbm_tb_s"] * 1e3, 1),
                        peak=d["peak"] if bound == "mfma" else HBM_PEAK_TBS * 1e3, unit="TFLOP/s" if bound == "mfma" else "GB/s",
                        frac=d["frac"] if bound == "mfma" else d["hbm_frac"], traffic=d["traffic"], avg_launch_ms=d["avg_launch_ms"],
                        launches=d["launches"], algorithmic_gflop_per_launch=d["algorithmic_gflop_per_launch"],
                        sclk_ghz=d["sclk_ghz"], frac_at_measured_clock=d["frac_at_measured_clock"],
                        mfma=dict(achieved_tflops=d["tflops"], peak_tflops=d["peak"], frac=d["frac"]),
                        hbm=dict(achieved_tb_s=d["hbm_tb_s"], peak_tb_s=HBM_PEAK_TBS, frac=d["hbm_frac"]),
                        lowest_frac_kernel=kernels[lowest]["kernel"], lowest_frac=kernels[lowest]["frac"],
                        mlp_kernels=kernels,
                        kernel_ms_per_step={nm: round(m / args.steps, 4) for nm, (_, m) in sorted(kern.items(), key=lambda kv: -kv[1][1])})
        if args.mode == "train":
            total_flops = 2.0 * fwd_macs * (m_c + m_f) + 2.0 * (fwd_macs + dgrad_macs) * (kept_c + kept_f)   # (executed: == algorithmic when dense)
            if eff_compact in ("recompute", "fused", "fused_compact"):
                total_flops += 2.0 * fwd_macs * (kept_c + kept_f)
            step_bytes = stash_b * (m_c + m_f) + (dgrad_b + wgrad_b) * (kept_c + kept_f)
            if eff_compact in ("fused", "fused_compact"):
                step_bytes = 20 * (m_c + m_f) + 40 * (kept_c + kept_f)
            if eff_compact == "fused_stash":
                step_bytes = (36 + 8 * (64 * Ln + 192)) * (m_c + m_f)
            if args.workload == "lego":
                workload = ("lego %dx%d synthetic views (BASELINE configs[%d]): %d rays/GPU/iter (%d over all GPUs), %d coarse + %d "
                            "fine samples, %dx%d coarse+fine nets, perturb, noise 0.2, Adam, full iteration"
                            % (side, side, 2 if (strong and side == 800) else 1, n, total_rays, nc, nf, Ln, Wd))
            else:
                workload = ("fern / LLFF %dx%d synthetic forward-facing views (BASELINE configs[3], config/fern.yml): NDC rays, near 0 / far 1, "
                            "%d rays/GPU/iter (%d over all GPUs), %d coarse + %d fine samples, %dx%d skip-%d nets with %d xyz frequencies, "
                            "perturb, noise 1.0, Adam, full iteration"
                            % (height, width, n, total_rays, nc, nf, Ln, Wd, cfg["skip_connect_every"], cfg["num_encoding_fn_xyz"]))
            metric = "train rays/sec"
        else:
            total_flops = 2.0 * fwd_macs * (m_c + m_f)
            step_bytes = stash_b * (m_c + m_f)
            workload = ("eval_nerf.py 360-degree render (BASELINE configs[4]): %dx%d poses, rows sharded over %d GPU(s) (%d rays/GPU/"
                        "pose), %d coarse + %d fine samples, %dx%d nets, perturb off, 8-bit cast on the device, one step = one pose"
                        % (side, side, world, n, nc, nf, Ln, Wd))
            metric = "eval rays/sec"
        sec = dt / args.steps

        def describe(prec):
            fmt, level = precision_level(prec)
            if level == 0:
                return "f32"
            pieces = ("two IEEE fp16 pieces per operand (hi + lo reproduces the fp32 value to 2^-24; per-sample block floating point against "
                      "fp16's range), three fp16 MFMAs per product block, f32 accumulate: ~3 x 2^-24 per product")
            what = {1: "forward", 2: "forward (backward + optimizer f32)", 3: "forward + data gradient (weight gradient + optimizer f32)",
                    4: "forward, data gradient and the hidden x hidden weight-gradient blocks (thin weight-gradient blocks + optimizer f32)"}[level]
            return "%s: %s on %s; fp32-equivalent FLOPs" % (fmt, what, pieces)
        dtype = describe(prec_f) if prec_c == prec_f else "coarse net: %s | fine net: %s" % (describe(prec_c), describe(prec_f))
        res = dict(metric=metric, value=round(total_rays * args.steps / dt, 2), unit="rays/s", n_gpus=world,
                   steps=args.steps, warmup=args.warmup, ms_per_step=round(sec * 1e3, 3),
                   ms_per_step_per_rank=[round(t / args.steps * 1e3, 3) for t in per_rank],
                   higher_is_better=True, scaling="strong" if strong else "weak", vs_baseline=None,
                   dtype=dtype, precision=args.precision, data="synthetic",
                   config=dict(workload=workload, rays_per_gpu=n, global_rays=total_rays, parallelism="dp%d" % world,
                               two_stream_step=bool(eng.overlap) if args.mode == "train" else None,
                               backend=("gloo(one-device test hook)" if one_device else "nccl(RCCL %s)" % rccl) if world > 1 else None),
                   # every launch of the timed region is bracketed by two HIP events on its stream (nerfhip_profile_enable): the
                   # per-kernel times below come from THIS run; measured cost of the events: none (27.31 vs 27.41 ms)
                   profiled_in_timed_region=not args.no_kernel_profile, host_issue_ms_per_step=round(dt_issue / args.steps * 1e3, 3),
                   unprofiled_rerun=unprofiled, lib_sources_sha16=lib_sources_sha16(),
                   step_tflops=round(total_flops / sec / 1e12, 2),
                   step_frac_of_fp32_mfma_peak=round(total_flops / sec / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                   step_algorithmic_hbm_tb_s=round(step_bytes / sec / 1e12, 3),
                   step_hbm_frac_of_8tb_s=round(step_bytes / sec / 1e12 / HBM_PEAK_TBS, 4),
                   final_loss=loss_host, roofline=roof)
        if args.mode == "train":
            res["backward"] = {None: "dense", "dense": "dense", "gather": "compacted", "recompute": "compacted, stash recomputed for the kept samples",
                               "fused": "fused (one persistent kernel per net: forward recomputed, data gradient, weight gradient; no stash, no d(pre-activation) images)",
                               "fused_compact": "fused, over the samples whose d(loss)/d(raw) row is non-zero",
                               "fused_stash": "fused over a register-image stash (one persistent kernel per net: data gradient + weight gradient; the forward leaves the chain's "
                                              "registers in HBM, 1.8 KB per sample point, instead of being recomputed)",
                               "auto": "auto (per net and step: dense / compacted / recomputed by the zero fraction of earlier steps)"}[args.compact if args.compact else eff_compact]
            if args.compact == "auto":
                res["backward_modes_used"] = dict(steps_dense_compacted_recomputed=eng.backward_modes_used, last_known_zero_fraction=eng._zero_frac)
        if kept is not None:
            frac = lambda kv, m: None if kv is None else round(1.0 - kv[0] / float(m), 4)  # noqa: E731
            res["zero_cotangent_fraction"] = dict(
                coarse=frac(kept["coarse"], m_c), fine=frac(kept["fine"], m_f),
                backward_sample_points=round(1.0 - (kept_c + kept_f) / float(m_c + m_f), 4),
                kept_sample_points=dict(coarse=kept_c, fine=kept_f), sample_points=dict(coarse=m_c, fine=m_f),
                what="sample points whose d(loss)/d(raw) row is exactly zero in the last timed step (relu(sigma + noise) off, or behind the "
                     "sample where the ray's transmittance reaches 0: nerf/volume_rendering_utils.py:38-42) and which the compacted backward "
                     "therefore dropped; the backward kernels' FLOPs, bytes and roofline fractions in this line count the KEPT sample points "
                     "only (executed work), step_tflops likewise")
        if world > 1:
            # self-diagnosis of the first real N-GPU run: the exchange next to the compute, per net
            res["multi_gpu"] = dict(nranks=world, rccl=rccl, ms_per_step_per_rank=res["ms_per_step_per_rank"],
                                    allreduce_ms_per_step=allreduce_ms,
                                    gradient_bytes_per_net=4 * mc.num_flat_params,
                                    note="allreduce_ms_per_step: one all-reduce of each net's flat gradient by itself, after the timed region "
                                         "(HIP events, mean of 20: TrainEngine.collective_times_ms) -- in the step the fine net's overlaps the "
                                         "coarse backward; SURVEY 5.8 expects ~55 us for 4.77 MB over xGMI.  ms_per_step_per_rank is compute + exchange")
        if world == 1 and not args.no_cpu_baseline:
            if args.mode == "train":
                res["cpu_baseline"] = cpu_baseline(wl=wl, cfg=cfg)
                if args.workload == "lego" and cfg == MODEL:
                    try:
                        res["dropin_route"] = dropin_route(dev)
                    except Exception as e:
                        res["dropin_route"] = dict(error=repr(e)[:200])
                try:
                    res["pytorch_rocm_reference"] = pytorch_rocm_reference(dev, wl=wl, cfg=cfg)
                    res["speedup_vs_pytorch_rocm_fwd_bwd"] = round(res["value"] / res["pytorch_rocm_reference"]["value"], 3)
                except Exception as e:  # the torch arm needs ~13 GB and must never take the bench line down
                    res["pytorch_rocm_reference"] = dict(error=repr(e)[:200])
                if args.workload == "lego" and cfg == MODEL and args.precision == "fp32" and not args.compact and not args.no_labelled_lines:
                    # Labelled lines, each measured by this very script in a child process (`value` above stays the dense fp32 headline):
                    # the same workload on the fp16-piece kernels (the arithmetic that holds the parity suite at the fp32 kernels' bounds:
                    # tests/test_gpu_fullsize.py); both with the compacted backward; BASELINE configs[3] (fern) and [4] (eval,
                    # one pose); the 4x128 nets the reference's scripts really build (train_nerf.py:117-134); and the trained regime
                    # (scripts/bench_trained.py: 2000 iterations on the teacher scene first -- where most cotangent rows are zero)
                    ref = res["pytorch_rocm_reference"].get("value")
                    res["labelled_lines"] = {}
                    short = ["--no-cpu-baseline", "--no-labelled-lines", "--steps", str(min(args.steps, 10)), "--warmup", "2"]
                    # (fp16-piece plans run the two-stream step by default -- +1.4 % dense, +7 % in the trained regime --, which makes a
                    # kernel's event-to-event time that of a GPU it shares: the f16x3_train line keeps ONE stream, so that its per-kernel
                    # times are the kernels' own, and the engine's default is the line next to it)
                    children = (("f16x3_train", ["--precision", "f16x3_train", "--overlap", "0", "--steps", str(args.steps), "--warmup", str(args.warmup)]),
                                ("f16x3_train_two_stream", ["--precision", "f16x3_train", "--steps", str(args.steps), "--warmup", str(args.warmup)]),
                                ("fp32_compact", ["--compact"]),
                                ("f16x3_train_compact", ["--precision", "f16x3_train", "--compact", "--overlap", "0"]),
                                ("fern_fp32", ["--workload", "fern"]),   # (4 x 64 fp32 nets: the fused one-kernel backward over the register-image stash is their default)
                                ("fern_fp32_recompute", ["--workload", "fern", "--compact", "fused"]),   # (... the same kernel recomputing its forward: no stash at all)
                                # (nets of hidden_size <= 128 run the two-stream step by default: the coarse backward shares the GPU with the fine pass,
                                # and a kernel's event-to-event time -- hence its fraction of a roofline -- is then that of HALF a GPU at times.  The
                                # same lines on ONE stream: what the kernels do when they have the chip to themselves)
                                ("fern_fp32_one_stream", ["--workload", "fern", "--overlap", "0"]),
                                ("fern_fp32_dense", ["--workload", "fern", "--compact", "dense"]),
                                ("fern_f16x3_train", ["--workload", "fern", "--precision", "f16x3_train"]),
                                ("4x128_fp32", ["--hidden", "128", "--layers", "4"]),
                                ("4x128_fp32_one_stream", ["--hidden", "128", "--layers", "4", "--overlap", "0"]),
                                ("4x128_f16x3_train", ["--hidden", "128", "--layers", "4", "--precision", "f16x3_train"]),
                                ("eval_fp32", ["--mode", "eval", "--steps", "1", "--warmup", "1"]),
                                ("eval_f16x3", ["--mode", "eval", "--precision", "f16x3", "--steps", "1", "--warmup", "1"]))
                    for label, extra in children:
                        try:
                            cmd = [sys.executable, os.path.abspath(__file__)] + short + ["--rays", str(args.rays)] + extra
                            out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
                            j = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
                            line = dict(metric=j["metric"], value=j["value"], unit=j["unit"], ms_per_step=j["ms_per_step"], steps=j["steps"],
                                        dtype=j["dtype"], workload=j["config"]["workload"], backward=j.get("backward"),
                                        step_tflops_executed=j["step_tflops"],
                                        mlp_kernels={k: dict(kernel=v["kernel"], ms_per_step=v["ms_per_step"], frac_of_own_mfma_roofline=v["frac"],
                                                             hbm_tb_s=v["hbm_tb_s"], hbm_frac=v["hbm_frac"])
                                                     for k, v in j["roofline"]["mlp_kernels"].items()},
                                        dominant_kernel=dict(kernel=j["roofline"]["kernel"], bound=j["roofline"]["bound"], frac=j["roofline"]["frac"],
                                                             traffic=j["roofline"].get("traffic")),
                                        two_stream_step=j["config"].get("two_stream_step"),
                                        command="python bench.py " + " ".join(extra))
                            if j.get("unprofiled_rerun"):   # (the same steps without the per-launch HIP events: what a 2-ms step pays for them)
                                line["ms_per_step_without_launch_events"] = j["unprofiled_rerun"]["ms_per_step"]
                            if "zero_cotangent_fraction" in j:
                                z = j["zero_cotangent_fraction"]
                                line["zero_cotangent_fraction"] = dict(coarse=z["coarse"], fine=z["fine"], backward_sample_points=z["backward_sample_points"])
                            if label.startswith(("f16x3", "fp32_compact")):
                                line["vs_fp32_line"] = round(j["value"] / res["value"], 3)
                                line["speedup_vs_pytorch_rocm_fwd_bwd"] = round(j["value"] / ref, 3) if ref else None
                            res["labelled_lines"][label] = line
                        except Exception as e:
                            res["labelled_lines"][label] = dict(error=repr(e)[:200])
                    try:
                        tmp = os.path.join(ROOT, "gpurun_out", "bench_trained_regime.json")
                        os.makedirs(os.path.dirname(tmp), exist_ok=True)
                        subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "bench_trained.py"), tmp, "--iters", "2000", "--steps",
                                        str(min(args.steps, 10)), "--warmup", "2"], capture_output=True, text=True, timeout=600)
                        t = json.load(open(tmp))
                        tr = dict(what="%s; students %s, %s samples, %d rays/step; trained %d iterations (%s) before the timed steps; full iterations "
                                       "(selection from resident views, forward, loss, backward, Adam, re-pack)"
                                       % (t["scene"], t["student"], t["samples"], t["rays_per_step"], t["pretrain_iters"], t["pretrain"]["arm"]),
                                  zero_cotangent_fraction_while_training=t["pretrain"]["zero_cotangent_fraction_at_iteration"],
                                  val_psnr_after_pretraining=t["pretrain"].get("val_psnr"), command="python scripts/bench_trained.py OUT.json")
                        for arm, v in t["arms"].items():
                            tr[arm] = dict(value=v["rays_per_s"], unit="rays/s", ms_per_step=v["ms_per_step"],
                                           zero_cotangent_fraction=v["zero_cotangent_fraction_last_step"],
                                           speedup_vs_pytorch_rocm_fwd_bwd=round(v["rays_per_s"] / ref, 3) if ref else None,
                                           grad_vs_dense_of_max=v.get("grad_vs_dense_of_max"), kernel_ms_per_step=v.get("kernel_ms_per_step"))
                            if arm.startswith("dropin_"):   # (the reference's own loop on the drop-in API: scripts/bench_trained.py part 3)
                                tr[arm].update(what=v.get("what"), backward_modes_at_the_end=v.get("backward_modes_at_the_end"))
                        res["labelled_lines"]["trained_regime"] = tr
                    except Exception as e:
                        res["labelled_lines"]["trained_regime"] = dict(error=repr(e)[:200])
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
