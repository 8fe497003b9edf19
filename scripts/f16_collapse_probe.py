"""Diagnostic (round 4): seed 2 of the 8x256 PSNR study collapsed (fine net renders a constant) in the engine_f16tr arm between
iterations 1000 and 1500 while the fp32 arms did not.  Re-runs that seed per arm with per-step checks -- loss, gradient norm /
absmax / finiteness, parameter absmax -- and reports the first anomalies.   python scripts/f16_collapse_probe.py SEED ITERS ARM[,ARM..] [LR]"""
import sys, os, json
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "scripts")):
    sys.path.insert(0, p)
import nerf_pytorch_amd as N  # noqa: E402
import psnr400 as P4  # noqa: E402
import psnr_arms as A  # noqa: E402

seed, iters, arms = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3].split(",")
lr0 = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-3
dev = torch.device("cuda", 0)
student = dict(num_layers=8, hidden_size=256, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)
poses, imgs, train, val = P4.teacher_dataset()
PREC = {"engine": "fp32", "engine_f16fwd": "f16x3_fwd", "engine_f16fd": "f16x3_fwd_dgrad", "engine_f16tr": "f16x3_train"}
for arm in arms:
    torch.manual_seed(seed)
    mc, mf = N.FlexibleNeRFModel(**student).to(dev), N.FlexibleNeRFModel(**student).to(dev)
    if PREC[arm] != "fp32":
        mc.set_training_precision(PREC[arm]); mf.set_training_precision(PREC[arm])
    eng = N.TrainEngine(mc, mf, A.NC, A.NF, perturb=True, white_background=True, noise_std=0.2, lr=lr0, seed=seed)
    stream = A.data_stream(poses, imgs, train, seed)
    opts = N.make_options(A.NC, A.NF, white_background=True)
    hist = []
    for i in range(1, iters + 1):
        ro, rd, tgt = next(stream)
        rays = N.pack_rays(ro, rd, opts)
        eng.forward_backward(rays, tgt)
        g = eng.grad
        nc = eng.nc_params
        rec = (i, float(eng.loss[0]), float(eng.loss[1]), float(g[:nc].norm()), float(g[nc:].norm()), float(g.abs().max()), bool(torch.isfinite(g).all()),
               float(mf.flat_params.abs().max()))
        hist.append(rec)
        eng.optimizer_step(N.TrainEngine.lr_at(i - 1, lr0=lr0))
        if i % 100 == 0 or not rec[6]:
            print(arm, "it %d loss c %.4f f %.4f |g_c| %.3e |g_f| %.3e gmax %.3e finite %s pmax %.2f" % rec, flush=True)
    h = np.array([[r[1], r[2], r[3], r[4], r[5]] for r in hist])
    med = np.median(h[:, 3])
    spikes = [hist[k] for k in range(len(hist)) if h[k, 3] > 20 * np.median(h[max(0, k - 50):k + 1, 3]) or h[k, 2] > 20 * np.median(h[max(0, k - 50):k + 1, 2])]
    print(arm, "gradient-norm spikes (> 20x the running median):", len(spikes))
    for r in spikes[:20]:
        print("   it %d loss c %.4f f %.4f |g_c| %.3e |g_f| %.3e gmax %.3e finite %s pmax %.2f" % r)
    jump = [k for k in range(1, len(hist)) if h[k, 1] > 3 * h[k - 1, 1] and h[k, 1] > 0.05]
    print(arm, "fine-loss jumps:", [(hist[k][0], round(h[k - 1, 1], 4), round(h[k, 1], 4)) for k in jump[:10]])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(hist, open(os.path.join(ROOT, "gpurun_out", "collapse_probe_%s_seed%d.json" % (arm, seed)), "w"))
