"""GPU-box measurement (round 6): the training step in the TRAINED regime, dense against compacted backward.

bench.py's default workload trains on uniform-noise images: after its three warm-up Adam steps the nets' sigma is positive everywhere and
no sample's cotangent is zero -- the compacted backward has nothing to drop there (its line says so: `zero_cotangent_fraction`).  A real
scene is mostly empty space: relu(sigma + noise) is off in front of and between the objects and the transmittance dies behind the first
surface (nerf/volume_rendering_utils.py:38-42), so most d(loss)/d(raw) rows are exactly zero (the 20 000-iteration soak of round 5
recorded 75-87 % on this scene).  This script

  1. trains 8x256 coarse + fine students for --iters iterations on the teacher scene of scripts/psnr400.py (lego views rendered from the
     reference's lego-lowres weights, white background, 4096 rays of a 400x400 view per step, 64 + 128 samples) with the engine that
     ships (f16x3_train plans, compacted backward: the fastest arm; the weights it leaves are what every arm below starts from),
  2. then times, for each arm in {fp32, f16x3_train} x {dense, compacted, recomputed (stash-free forward + forward again for the kept samples)}: --steps full training iterations (ray selection from a
     resident view, forward, loss, backward, Adam, re-pack) after --warmup, HIP-event bracketed per kernel, on the same data stream,
  3. (and the same regime through the reference's own loop on the drop-in API -- run_one_iter_of_nerf, loss.backward(), torch.optim.Adam --
     dense and with set_backward_compaction("auto") on the models: the arms dropin_*),
  4. and records per arm: rays/s, ms/step, the zero-cotangent fraction of the last timed step per net (compacted arms: read from the
     library; dense arms: the same step's d(raw) rows counted with torch), per-kernel ms/step, and -- compacted vs dense, same weights,
     same rays, same draws -- the largest gradient difference of max|g| per net.

    python scripts/bench_trained.py OUT.json [--iters 2000] [--steps 20] [--warmup 3] [--lr 1e-3]
"""
import argparse
import copy
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "scripts")):
    sys.path.insert(0, p)
import nerf_pytorch_amd as N  # noqa: E402
import nerf_pytorch_amd._lib as L  # noqa: E402
import psnr400 as P4  # noqa: E402
from psnr_arms import data_stream  # noqa: E402

dev = torch.device("cuda", 0)
NC, NF, RAYS = 64, 128, 4096
STUDENT = dict(num_layers=8, hidden_size=256, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)


OVERLAP = None  # (--overlap: the engine's two-stream step; None = its default for the net width)


def make_engine(state_c, state_f, precision, compact, lr, seed):
    """compact: False / True / "recompute" / "fused" / "fused_compact" (set on both models) or "auto" (TrainEngine(backward="auto"))."""
    mc, mf = N.FlexibleNeRFModel(**STUDENT), N.FlexibleNeRFModel(**STUDENT)
    if state_c is not None:
        mc.load_state_dict(state_c)
        mf.load_state_dict(state_f)
    mc, mf = mc.to(dev), mf.to(dev)
    if precision != "fp32":
        mc.set_training_precision(precision)
        mf.set_training_precision(precision)
    if compact != "auto":
        mc.set_backward_compaction(compact)
        mf.set_backward_compaction(compact)
    eng = N.TrainEngine(mc, mf, NC, NF, perturb=True, white_background=True, noise_std=0.2, lr=lr, seed=seed,
                        backward="auto" if compact == "auto" else None, overlap=OVERLAP)
    return mc, mf, eng


def timed(eng, stream, opts, steps, warmup, lr):
    lib = L.get_lib()
    batches = [next(stream) for _ in range(warmup + steps)]
    for ro, rd, tgt in batches[:warmup]:
        eng.step(N.pack_rays(ro, rd, opts), tgt, lr=lr)
    lib.profile_reserve(96 * steps)
    torch.cuda.synchronize()
    lib.profile_enable(1)
    t0 = time.perf_counter()
    for ro, rd, tgt in batches[warmup:]:
        eng.step(N.pack_rays(ro, rd, opts), tgt, lr=lr)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    lib.profile_enable(0)
    buf = C.create_string_buffer(1 << 16)
    lib.profile_report(buf, len(buf))
    kern = {}
    for line in buf.value.decode().splitlines():
        name, cnt, ms = line.rsplit(" ", 2)
        kern[name.strip("()")] = round(float(ms) / steps, 4)
    return dt / steps, dict(sorted(kern.items(), key=lambda kv: -kv[1])[:12])


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--iters", type=int, default=2000)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--save-weights", default=None, help="keep the trained weights (a torch file) for later calls")
    ap.add_argument("--load-weights", default=None, help="skip the training: start from a file written by --save-weights (profiler runs)")
    ap.add_argument("--hidden", type=int, default=256, help="hidden_size of the students (128 with --layers 4: the nets train_nerf.py:117-134 really builds)")
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--overlap", type=int, default=-1, help="1: two-stream step (coarse backward next to the fine pass), 0: one stream, -1: the engine's default")
    ap.add_argument("--arms", default=None, help="comma-separated subset of the arms, e.g. f16x3_train_compacted (default: all six)")
    a = ap.parse_args()
    OVERLAP = None if a.overlap < 0 else bool(a.overlap)
    STUDENT.update(num_layers=a.layers, hidden_size=a.hidden)
    lib = L.get_lib()
    poses, imgs, train, val = P4.teacher_dataset()
    opts = N.make_options(NC, NF, white_background=True)
    res = dict(scene="teacher scene of scripts/psnr400.py: 400x400 lego views rendered from the reference's lego-lowres weights, white background",
               student="%dx%d coarse + fine" % (a.layers, a.hidden), rays_per_step=RAYS, samples="64 + 128", pretrain_iters=a.iters, lr0=a.lr, steps=a.steps,
               warmup=a.warmup, lib_version=lib.version(), two_stream_step=OVERLAP, arms={})
    if a.load_weights:
        saved = torch.load(a.load_weights)
        state_c, state_f = saved["coarse"], saved["fine"]
        res["pretrain"] = saved["pretrain"]
    else:
        # 1. the trained regime
        torch.manual_seed(a.seed)
        mc, mf, eng = make_engine(None, None, "f16x3_train", True, a.lr, a.seed)
        stream = data_stream(poses, imgs, train, a.seed)
        t0 = time.perf_counter()
        fracs = {}
        for i in range(1, a.iters + 1):
            ro, rd, tgt = next(stream)
            eng.step(N.pack_rays(ro, rd, opts), tgt, lr=N.TrainEngine.lr_at(i - 1, lr0=a.lr))
            if i in (1, 10, 100, 500, 1000, 2000, 5000, 10000, 20000) or i == a.iters:
                k = eng.backward_sample_counts()
                fracs[i] = {n: round(1.0 - v[0] / v[1], 4) for n, v in k.items()}
                print("pretrain", i, fracs[i], flush=True)
        torch.cuda.synchronize()
        res["pretrain"] = dict(arm="f16x3_train, compacted backward", wall_s=round(time.perf_counter() - t0, 2), zero_cotangent_fraction_at_iteration=fracs,
                               train_psnr_last=float(N.TrainEngine.psnr(float(eng.loss[2]))))
        vals = P4.validate_hip(mc, mf, poses, imgs, val[:2])
        res["pretrain"]["val_psnr"] = P4.psnr(sum(c for c, _ in vals) / len(vals) + sum(f for _, f in vals) / len(vals))
        state_c, state_f = copy.deepcopy(mc.state_dict()), copy.deepcopy(mf.state_dict())
        del eng, mc, mf
        torch.cuda.empty_cache()
        if a.save_weights:
            torch.save(dict(coarse=state_c, fine=state_f, pretrain=res["pretrain"]), a.save_weights)
    # 2. the arms, all from the same weights on the same stream
    lr = N.TrainEngine.lr_at(a.iters, lr0=a.lr)
    grads = {}
    fused_ok = N.FlexibleNeRFModel(**STUDENT).to(dev).fused_backward_available()   # (fp32 students of hidden_size <= 64: csrc/mlp64r.hip)
    for prec in ("fp32", "f16x3_train"):
        for compact in (False, True, "recompute", "fused", "fused_compact", "fused_stash", "auto"):
            arm = "%s_%s" % (prec, {False: "dense", True: "compacted", "recompute": "recomputed", "auto": "auto"}.get(compact, compact))
            if a.arms and arm not in a.arms.split(","):
                continue
            if compact in ("fused", "fused_compact", "fused_stash") and not (fused_ok and prec == "fp32"):
                continue
            mc, mf, eng = make_engine(state_c, state_f, prec, compact, a.lr, a.seed + 7)
            # one step on a fixed batch first: the gradient this arm computes from the common weights (compacted vs dense below)
            s0 = data_stream(poses, imgs, train, 999)
            ro, rd, tgt = next(s0)
            eng.forward_backward(N.pack_rays(ro, rd, opts), tgt)
            torch.cuda.synchronize()
            grads[arm] = eng.grad.clone()
            ms, kern = timed(eng, data_stream(poses, imgs, train, 4242), opts, a.steps, a.warmup, lr)
            kept = eng.backward_sample_counts()
            zf = {n: (None if v is None else round(1.0 - v[0] / v[1], 4)) for n, v in kept.items()}
            if compact == "auto":
                zf["steps_dense_compacted_recomputed_fused_fusedcompact_fusedstash"] = eng.backward_modes_used
            res["arms"][arm] = dict(rays_per_s=round(RAYS / ms, 1), ms_per_step=round(ms * 1e3, 3), zero_cotangent_fraction_last_step=zf,
                                    kernel_ms_per_step=kern, final_loss=[float(v) for v in eng.loss.cpu()])
            print(arm, res["arms"][arm]["rays_per_s"], res["arms"][arm]["ms_per_step"], zf, flush=True)
            del eng, mc, mf
            torch.cuda.empty_cache()
            json.dump(res, open(a.out, "w"), indent=1)
    n0 = None
    for prec in ("fp32", "f16x3_train"):
        for kind in ("_compacted", "_recomputed", "_fused", "_fused_compact", "_fused_stash", "_auto"):
            if prec + "_dense" not in grads or prec + kind not in grads:
                continue
            d, c = grads[prec + "_dense"], grads[prec + kind]
            n0 = d.numel() // 2
            res["arms"][prec + kind]["grad_vs_dense_of_max"] = dict(
                coarse=float((c[:n0] - d[:n0]).abs().max() / d[:n0].abs().max()), fine=float((c[n0:] - d[n0:]).abs().max() / d[n0:].abs().max()))
            res["arms"][prec + kind]["speedup_vs_dense"] = round(res["arms"][prec + "_dense"]["ms_per_step"] / res["arms"][prec + kind]["ms_per_step"], 3)
    # 3. the same regime through the REFERENCE'S OWN LOOP on the drop-in API (import swap; no TrainEngine):
    # run_one_iter_of_nerf, img2mse, loss.backward(), torch.optim.Adam.step() (train_nerf.py:226-261) -- dense, and with
    # set_backward_compaction("auto") on the two models (the policy of the engine's backward="auto", per net and forward pass)
    ex = N.get_embedding_function(STUDENT["num_encoding_fn_xyz"], True, True)
    ed = N.get_embedding_function(STUDENT["num_encoding_fn_dir"], True, True)
    for prec in ("fp32", "f16x3_train"):
        for mode in ("dense", "auto"):
            arm = "dropin_%s_%s" % (prec, mode)
            if a.arms and arm not in a.arms.split(","):
                continue
            mc, mf, _eng = make_engine(state_c, state_f, prec, False, a.lr, a.seed + 7)
            del _eng
            if mode == "auto":
                mc.set_backward_compaction("auto"), mf.set_backward_compaction("auto")
            optim = torch.optim.Adam(list(mc.parameters()) + list(mf.parameters()), lr=lr)
            stream = data_stream(poses, imgs, train, 4242)
            warm = 60 if mode == "auto" else a.warmup    # (the first statistics must have come back, and the probe cadence settled)
            batches = [next(stream) for _ in range(a.steps)]

            def loop_step(ro, rd, tgt):
                out = N.run_one_iter_of_nerf(400, 400, 1.0, mc, mf, ro, rd, opts, mode="train", encode_position_fn=ex, encode_direction_fn=ed)
                loss = N.img2mse(out[0], tgt[..., :3]) + N.img2mse(out[3], tgt[..., :3])
                loss.backward()
                optim.step()
                optim.zero_grad()

            for _ in range(warm):
                loop_step(*next(stream))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for ro, rd, tgt in batches:
                loop_step(ro, rd, tgt)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / a.steps
            res["arms"][arm] = dict(rays_per_s=round(RAYS / ms, 1), ms_per_step=round(ms * 1e3, 3),
                                    zero_cotangent_fraction_last_step=dict(coarse=getattr(mc, "_auto_frac", None), fine=getattr(mf, "_auto_frac", None)),
                                    backward_modes_at_the_end=[mc.backward_compaction, mf.backward_compaction],
                                    what="run_one_iter_of_nerf + img2mse + loss.backward() + torch.optim.Adam.step(): the reference's loop body on the drop-in API")
            print(arm, res["arms"][arm]["rays_per_s"], res["arms"][arm]["ms_per_step"], res["arms"][arm]["backward_modes_at_the_end"], flush=True)
            del mc, mf, optim
            torch.cuda.empty_cache()
            json.dump(res, open(a.out, "w"), indent=1)
    json.dump(res, open(a.out, "w"), indent=1)
    print(json.dumps({k: (v["rays_per_s"], v["ms_per_step"], v["zero_cotangent_fraction_last_step"]) for k, v in res["arms"].items()}))
