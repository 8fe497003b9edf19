"""GPU-box experiment (round 5): does the fp16-piece training step survive a LONG run, on the kernels that ship?

Round 4's evidence for `f16x3_train` was 2 000 iterations, and the one range bug that mattered (NaN gradients at iteration 1257,
scripts/f16_collapse_probe.py) was found by a PSNR run, not by a parity test.  The reference trains 100 k - 250 k iterations
(pretrained/*/checkpoint*.ckpt); late-training statistics -- sigma saturated behind surfaces, cotangents ~1e-12 on most rays --
are what per-sample exponents, region maxima and the 2^13 target have never seen.  This script trains the SAME student twice per seed,

    engine        TrainEngine, fp32 plans (the reference's arithmetic; what bench.py's default line times)
    engine_f16tr  TrainEngine, NERFHIP_PRECISION_F16X3_TRAIN plans: every GEMM of the step on fp16 pieces (mlp_f16w.hip, wgrad_f16.hip)

for --iters iterations at 8x256 on the teacher scene of scripts/psnr400.py (same initial weights, views, pixels and Philox draws in both
arms), and records

  * validation PSNR (train_nerf.py:339-347's protocol: whole held-out views, -10 log10(coarse_mse + fine_mse), :258-260) every --check,
  * every --diag iterations, on 256 rays of the step just taken and the fine net (192 samples per ray: 49,152 sample points):
      - NaN / Inf check of the step's flat gradient (both nets) and of the loss;
      - the per-sample exponent range: floor(log2(max_u |h_k[sample, u]|)) of every stashed activation and of every d(pre-activation)
        image, min / max over the samples, recomputed with torch ops from the same weights (the kernels derive a sample's exponent from
        exactly this maximum: mlp_f16w.hip gemm_w / renorm_convert), and how many samples are all-zero (neutral exponent);
      - (engine_f16tr) the region maxima the kernels themselves recorded (the 64 words behind the stash / the backward scratch:
        note_region, e - 256 = log2 of the bound a region's values sit under), and the distance between the fp16-piece gradient of
        that sub-batch (nerfhip_mlp_fwd / nerfhip_mlp_bwd on the training plan) and torch's fp32 gradient, per tensor, of max|g| --
        on the samples that pass the ReLU-margin filter of tests/tolerances.py (the first soak compared unfiltered: its distances
        are either 2-3e-6 or one flipped ReLU branch).

    python scripts/psnr_soak.py SEED ITERS OUT.json [--arms engine,engine_f16tr] [--lr 1e-3] [--check 2000] [--diag 1000]
"""
import argparse
import ctypes as C
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
import nerf_pytorch_amd._lib as L  # noqa: E402
import psnr400 as P4  # noqa: E402  (teacher dataset, validation renders)
from psnr_arms import data_stream  # noqa: E402
from bench import lib_sources_sha16  # noqa: E402  (the fingerprint of the kernel sources the loaded library was built from)

dev = torch.device("cuda", 0)
H = W = 400
FOCAL = P4.FOCAL
NC, NF, RAYS = 64, 128, 4096
DIAG_RAYS = 256


def _exp_range(v):
    """v: (samples, units) -> min / max over the samples of floor(log2(max_u |v|)), and the all-zero samples."""
    m = v.detach().abs().amax(dim=1)
    nz = m > 0
    if not bool(nz.any()):
        return dict(min=None, max=None, zero_samples=int(m.numel()))
    e = torch.floor(torch.log2(m[nz]))
    return dict(min=int(e.min()), max=int(e.max()), zero_samples=int((~nz).sum()))


def _instrumented(params, x, cfg):
    """FlexibleNeRFModel.forward (nerf/models.py:233-256) in torch ops, keeping what the training kernels stash: the inputs of every
    gemm (H_0 .. H_{L-1}, FEAT, DIRH) and, with retain_grad, the pre-activations whose gradients are the d(pre-activation) images."""
    F = torch.nn.functional
    dx, dd = O.model_dims(cfg)
    xyz, dirs = x[..., :dx], x[..., dx:]
    acts, pres = {}, {}
    h = F.linear(xyz, params["layer1.weight"], params["layer1.bias"])
    pres["P0"] = h
    for i in range(cfg["num_layers"] - 1):
        acts["H%d" % i] = h
        hin = torch.cat((h, xyz), dim=-1) if O.is_skip_layer(i, cfg) else h
        pre = F.linear(hin, params["layers_xyz.%d.weight" % i], params["layers_xyz.%d.bias" % i])
        pres["P%d" % (i + 1)] = pre
        h = F.relu(pre)
    acts["H%d" % (cfg["num_layers"] - 1)] = h
    pre = F.linear(h, params["fc_feat.weight"], params["fc_feat.bias"])
    pres["PFEAT"] = pre
    feat = F.relu(pre)
    acts["FEAT"] = feat
    alpha = F.linear(h, params["fc_alpha.weight"], params["fc_alpha.bias"])
    pre = F.linear(torch.cat((feat, dirs), dim=-1), params["layers_dir.0.weight"], params["layers_dir.0.bias"])
    pres["PDIR"] = pre
    dh = F.relu(pre)
    acts["DIRH"] = dh
    rgb = F.linear(dh, params["fc_rgb.weight"], params["fc_rgb.bias"])
    for t in pres.values():
        t.retain_grad()
    return torch.cat((rgb, alpha), dim=-1), acts, pres  # (raw: the caller differentiates w.r.t. it with torch.autograd.grad)


def diagnose(eng, mf, rays, tgt, student, f16):
    lib = L.get_lib()
    out = dict(grad_finite=bool(torch.isfinite(eng.grad).all()), loss_finite=bool(torch.isfinite(eng.loss).all()),
               grad_absmax=float(eng.grad.abs().max()))
    n = rays.shape[0]
    off, nbytes = C.c_int64(), C.c_int64()
    lib.render_workspace_region(eng.mc._plan, eng.mf._plan, C.byref(eng.cfg), n, 1, b"z_fine", C.byref(off), C.byref(nbytes))
    z = eng._ws.view(torch.uint8)[off.value:off.value + nbytes.value].view(torch.float32).reshape(n, NC + NF)[:DIAG_RAYS].clone()
    r = rays[:DIAG_RAYS]
    pts = (r[:, None, :3] + r[:, None, 3:6] * z[..., None]).reshape(-1, 3)
    emb = O.positional_encoding(pts, student["num_encoding_fn_xyz"], True, True)
    dirs = r[:, None, -3:].expand(DIAG_RAYS, NC + NF, 3).reshape(-1, 3)
    x = torch.cat((emb, O.positional_encoding(dirs, student["num_encoding_fn_dir"], True, True)), dim=-1).contiguous()
    params = {k: v.detach().clone().requires_grad_(True) for k, v in mf.state_dict().items()}
    raw, acts, pres = _instrumented(params, x, student)
    rgb = O.volume_render(raw.reshape(DIAG_RAYS, NC + NF, 4), z, r[:, 3:6], 0.0, None, white_background=True)[0]
    # (the step's loss is a mean over 4096 rays: the sub-batch's cotangents get the same 1 / (3 * 4096))
    loss = ((rgb - tgt[:DIAG_RAYS, :3]) ** 2).sum() / (3.0 * n)
    g_raw, = torch.autograd.grad(loss, raw, retain_graph=True)
    raw.backward(g_raw, retain_graph=True)          # every sample: the exponent statistics
    out["activation_exponents"] = {k: _exp_range(v) for k, v in acts.items()}
    out["dpre_exponents"] = {k: _exp_range(v.grad) for k, v in pres.items()}
    out["raw_cotangent_exponents"] = _exp_range(g_raw)
    out["sigma_raw_max"] = float(raw[:, 3].max())
    # The gradient comparison below runs on the samples whose ReLU branches round-off cannot decide (tests/tolerances.py `relu_margin`:
    # no ReLU input within 1e-5, relative, of zero; layer1 has no activation): on a 49,152-sample batch ONE flipped branch moves a row of a
    # weight gradient by 1e-4 ... 5e-3 of max|g| (the unfiltered distances of the first soak, profiles/r05_psnr_soak.txt, are bimodal:
    # 2-3e-6 or a flip), for any two fp32-grade evaluations.
    relu_in = torch.cat([v.detach().abs() for k, v in pres.items() if k != "P0"], dim=1)
    keep = (relu_in.amin(dim=1) / (relu_in.amax(dim=1) + 1e-30)) > 1e-5
    out["relu_filter_kept_fraction"] = float(keep.float().mean())
    for v in params.values():
        v.grad = None
    g_keep = (g_raw * keep[:, None].float()).contiguous()
    raw.backward(g_keep)
    if f16:
        m = x.shape[0]
        plan, packed = mf._plan, mf._packed(True)
        y = torch.empty((m, 4), dtype=torch.float32, device=dev)
        stash = torch.empty(lib.plan_stash_bytes(plan, m) // 4, dtype=torch.float32, device=dev)
        sb = lib.plan_bwd_scratch_bytes(plan, m)
        scratch = torch.empty(sb // 4 + 1, dtype=torch.float32, device=dev)
        gflat = torch.empty(mf.num_flat_params, dtype=torch.float32, device=dev)
        g = g_keep.detach()
        with L.launch_on(x, y, packed, stash, scratch, gflat, g) as st:
            lib.mlp_fwd(plan, packed.data_ptr(), x.data_ptr(), m, y.data_ptr(), stash.data_ptr(), st)
            lib.mlp_bwd(plan, packed.data_ptr(), g.data_ptr(), m, stash.data_ptr(), scratch.data_ptr(), sb, gflat.data_ptr(), st)
        fw = stash[-64:].view(torch.int32).cpu().numpy()
        bw = scratch[:sb // 4][-64:].view(torch.int32).cpu().numpy()
        out["kernel_region_bound_log2"] = dict(stash=[int(v) - 256 for v in fw if v != 0], scratch=[int(v) - 256 for v in bw if v != 0])
        out["kernel_raw_vs_torch"] = float((y - raw.detach()).abs().max() / (raw.detach().abs().max() + 1e-30))
        out["kernel_grad_finite"] = bool(torch.isfinite(gflat).all())
        worst = 0.0
        for (name, o, rows, cols), gk in zip(mf._layout, mf._split_flat(gflat)):
            ref = params[name].grad
            worst = max(worst, float((gk - ref).abs().max() / (ref.abs().max() + 1e-30)))
        out["kernel_grad_vs_torch_worst_rel"] = worst
    return out


def run(arm, seed, iters, check, diag_every, student, lr0, poses, imgs, train, views, compact=False):
    P4.STUDENT.clear()
    P4.STUDENT.update(student)
    torch.manual_seed(seed)
    mc, mf = N.FlexibleNeRFModel(**student), N.FlexibleNeRFModel(**student)
    mc, mf = mc.to(dev), mf.to(dev)
    f16 = arm == "engine_f16tr"
    if f16:
        mc.set_training_precision("f16x3_train")
        mf.set_training_precision("f16x3_train")
    # (round 6: the compacted backward changes the summation order of every weight gradient -- its own long run)
    mode = {"compact": True, None: False, False: False}.get(compact, compact)   # ("recompute" / "fused" / "fused_compact" / "fused_stash": by name)
    mc.set_backward_compaction(mode)
    mf.set_backward_compaction(mode)
    eng = N.TrainEngine(mc, mf, NC, NF, perturb=True, white_background=True, noise_std=0.2, lr=lr0, seed=seed)
    opts = N.make_options(NC, NF, white_background=True)
    stream = data_stream(poses, imgs, train, seed)
    hist, diags, losses = {}, {}, []
    t_train, t_mark = 0.0, time.perf_counter()
    for i in range(1, iters + 1):
        ro, rd, tgt = next(stream)
        rays = N.pack_rays(ro, rd, opts)
        loss3 = eng.step(rays, tgt, lr=N.TrainEngine.lr_at(i - 1, lr0=lr0))
        losses.append(loss3[2:3].clone())
        if i % diag_every == 0 or i in check:
            torch.cuda.synchronize()
            t_train += time.perf_counter() - t_mark
            if i % diag_every == 0:
                kept = eng.backward_sample_counts() if compact else None
                diags[i] = diagnose(eng, mf, rays, tgt, student, f16)
                d = diags[i]
                if kept is not None:
                    d["zero_cotangent_fraction"] = {k: round(1.0 - v[0] / v[1], 4) for k, v in kept.items() if v}
                print(arm, seed, i, "diag", dict(finite=(d["grad_finite"], d["loss_finite"]), absmax=d["grad_absmax"],
                                                 H=d["activation_exponents"]["H%d" % (student["num_layers"] - 1)], P1=d["dpre_exponents"]["P1"],
                                                 k=d.get("kernel_grad_vs_torch_worst_rel")), flush=True)
            if i in check:
                recent = torch.cat(losses[-50:]).cpu().numpy()
                vals = P4.validate_hip(mc, mf, poses, imgs, views)
                vc, vf = float(np.mean([a for a, _ in vals])), float(np.mean([b for _, b in vals]))
                hist[i] = dict(train_psnr=P4.psnr(float(np.mean(recent))), val_psnr=P4.psnr(vc + vf), val_psnr_fine=P4.psnr(vf),
                               val_psnr_coarse=P4.psnr(vc), train_wall_s=round(t_train, 2))
                print(arm, seed, i, hist[i], flush=True)
            losses = losses[-50:]
            t_mark = time.perf_counter()
    return dict(checkpoints=hist, diagnostics=diags)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("seed", type=int)
    ap.add_argument("iters", type=int)
    ap.add_argument("out")
    ap.add_argument("--arms", default="engine_f16tr,engine")
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--check", type=int, default=2000)
    ap.add_argument("--diag", type=int, default=1000)
    ap.add_argument("--compact", nargs="?", const="compact", default=None, choices=("compact", "recompute", "fused", "fused_compact", "fused_stash"),
                    help="both arms with the compacted backward (set_backward_compaction(True)); `recompute`: its stash-recomputing form")
    a = ap.parse_args()
    student = dict(num_layers=a.layers, hidden_size=a.hidden, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)
    check = list(range(a.check, a.iters + 1, a.check)) or [a.iters]
    poses, imgs, train, val = P4.teacher_dataset()
    views = val[:P4.VAL_PER_CHECK]
    res = dict(seed=a.seed, iters=a.iters, lr0=a.lr, student="%dx%d" % (a.layers, a.hidden), rays_per_iter=RAYS, image="%dx%d" % (H, W),
               diag_rays=DIAG_RAYS, backward={None: "dense", "compact": "compacted", "recompute": "recomputed"}.get(a.compact, a.compact), lib_sources_sha16=lib_sources_sha16(),
               lib_version=L.get_lib().version(), arms={})
    for arm in a.arms.split(","):
        res["arms"][arm] = run(arm, a.seed, a.iters, check, a.diag, student, a.lr, poses, imgs, train, views, compact=a.compact)
        json.dump(res, open(a.out, "w"), indent=1)
