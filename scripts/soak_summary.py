"""profiles/r05_psnr_soak.txt from the runs of scripts/psnr_soak.py (gpurun_out/soak/*.json, copied to profiles/r05_psnr_soak_runs/):
    python scripts/soak_summary.py profiles/r05_psnr_soak_runs/soak_seed1.json profiles/r05_psnr_soak_runs/soak_seed2.json [--short FILE]"""
import json
import math
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
short = sys.argv[sys.argv.index("--short") + 1] if "--short" in sys.argv else None
if short:
    args.remove(short)
runs = [json.load(open(a)) for a in args]
P = print
P("# scripts/psnr_soak.py: %s students, %s, %d rays/iter, lr %g x 0.1^(i/250000), %d iterations, seeds %s" %
  (runs[0]["student"], runs[0]["image"], runs[0]["rays_per_iter"], runs[0]["lr0"], runs[0]["iters"], [r["seed"] for r in runs]))
P("# arms: engine = TrainEngine on fp32 plans; engine_f16tr = the same engine, same initial weights / views / pixels / Philox draws, every GEMM")
P("# of the step on fp16 pieces (NERFHIP_PRECISION_F16X3_TRAIN: k_mlp_fwd_f16x3w, k_mlp_dgrad_f16x3w, k_wgrad_f16x3 + guests).")
P("# validation PSNR = -10 log10(coarse_mse + fine_mse) on 3 whole held-out 400x400 views (train_nerf.py:258-260, :339-347), fp32 inference for both arms")
P()
P("%-7s" % "iter" + "".join("  seed %d: fp32   f16x3   delta " % r["seed"] for r in runs) + "   mean delta")
its = sorted(runs[0]["arms"]["engine"]["checkpoints"], key=int)
last = []
for i in its:
    row, ds = "%-7s" % i, []
    for r in runs:
        e, f = r["arms"]["engine"]["checkpoints"][i]["val_psnr"], r["arms"]["engine_f16tr"]["checkpoints"][i]["val_psnr"]
        ds.append(f - e)
        row += "          %6.3f  %6.3f  %+6.3f" % (e, f, f - e)
    P(row + "     %+6.3f" % (sum(ds) / len(ds)))
    last = ds
n = len(last)
mean = sum(last) / n
if n > 1:
    sd = math.sqrt(sum((d - mean) ** 2 for d in last) / (n - 1))
    t = {2: 12.706, 3: 4.303, 4: 3.182, 5: 2.776, 6: 2.571, 7: 2.447, 8: 2.365}.get(n, 2.0)  # (two-sided 95 %, n - 1 degrees of freedom)
    P("\nf16x3 - fp32 at iteration %s: mean %+.3f dB, 95 %% interval [%+.2f, %+.2f] (Student t, n = %d pairs);" %
      (its[-1], mean, mean - t * sd / math.sqrt(n), mean + t * sd / math.sqrt(n), n))
    alld = [r["arms"]["engine_f16tr"]["checkpoints"][i]["val_psnr"] - r["arms"]["engine"]["checkpoints"][i]["val_psnr"] for r in runs for i in its]
    P("over all %d checkpoint pairs: mean %+.3f dB, largest |delta| %.3f dB, sign + %d / - %d" %
      (len(alld), sum(alld) / len(alld), max(abs(d) for d in alld), sum(d > 0 for d in alld), sum(d < 0 for d in alld)))
P("training wall time for %s iterations (validation and diagnostics excluded): fp32 %s s, f16x3 %s s" %
  (its[-1], [r["arms"]["engine"]["checkpoints"][its[-1]]["train_wall_s"] for r in runs], [r["arms"]["engine_f16tr"]["checkpoints"][its[-1]]["train_wall_s"] for r in runs]))
P()
P("## range diagnostics, every 1000 iterations, fine net, 256 rays x 192 samples of the step just taken (f16x3 arm)")
for r in runs:
    dg = r["arms"]["engine_f16tr"]["diagnostics"]
    ks = sorted(dg, key=int)
    fin = all(dg[k]["grad_finite"] and dg[k]["loss_finite"] and dg[k]["kernel_grad_finite"] for k in ks)
    fin32 = all(v["grad_finite"] and v["loss_finite"] for v in r["arms"]["engine"]["diagnostics"].values())
    P("seed %d: %d diagnostics; every step's flat gradient (2 x 595,844 entries) and loss finite: %s (fp32 arm: %s)" % (r["seed"], len(ks), fin, fin32))
    amin = min(v["min"] for k in ks for v in dg[k]["activation_exponents"].values())
    amax = max(v["max"] for k in ks for v in dg[k]["activation_exponents"].values())
    dmin = min(v["min"] for k in ks for v in dg[k]["dpre_exponents"].values() if v["min"] is not None)
    dmax = max(v["max"] for k in ks for v in dg[k]["dpre_exponents"].values() if v["max"] is not None)
    zs = [max(v["zero_samples"] for v in dg[k]["dpre_exponents"].values()) for k in ks]
    P("  per-sample floor(log2 max|.|): stashed activations %d .. %d; d(pre-activation) images %d .. %d (all-zero samples: %d .. %d of 49,152)" %
      (amin, amax, dmin, dmax, min(zs), max(zs)))
    P("  largest raw sigma seen: %.0f; |flat gradient| max %.2e .. %.2e" % (max(dg[k]["sigma_raw_max"] for k in ks), min(dg[k]["grad_absmax"] for k in ks), max(dg[k]["grad_absmax"] for k in ks)))
    sb = [v for k in ks for v in dg[k]["kernel_region_bound_log2"]["stash"]]
    gb = [v for k in ks for v in dg[k]["kernel_region_bound_log2"]["scratch"]]
    P("  region bounds recorded by the kernels (log2): stash regions %d .. %d, d(pre-activation) regions %d .. %d" % (min(sb), max(sb), min(gb), max(gb)))
    P("  forward on fp16 pieces vs torch fp32, max over the sub-batch, of max|raw|: %.1e .. %.1e" % (min(dg[k]["kernel_raw_vs_torch"] for k in ks), max(dg[k]["kernel_raw_vs_torch"] for k in ks)))
    d = [dg[k]["kernel_grad_vs_torch_worst_rel"] for k in ks]
    filt = "relu_filter_kept_fraction" in dg[ks[0]]
    P("  fp16-piece gradient vs torch's fp32 gradient on that sub-batch, worst tensor, of max|g|, %s: median %.1e, max %.1e; per diagnostic: %s" %
      ("on the samples that pass the ReLU-margin filter (kept %.2f .. %.2f)" % (min(dg[k]["relu_filter_kept_fraction"] for k in ks),
                                                                               max(dg[k]["relu_filter_kept_fraction"] for k in ks)) if filt
       else "UNFILTERED (the first two seeds ran before the diagnostic filtered: 2-9e-6, or one flipped ReLU branch)",
       sorted(d)[len(d) // 2], max(d), " ".join("%.0e" % v for v in d)))
if short:
    s = json.load(open(short))
    dg = s["arms"]["engine_f16tr"]["diagnostics"]
    ks = sorted(dg, key=int)
    d = [dg[k]["kernel_grad_vs_torch_worst_rel"] for k in ks]
    P("\n## the same comparison on the samples that pass the ReLU-margin filter (tests/tolerances.py relu_margin = 1e-5), seed %d, %d iterations, every 250" % (s["seed"], s["iters"]))
    P("kept fraction %.3f .. %.3f; worst tensor, of max|g|: %s   (max %.1e)" %
      (min(dg[k]["relu_filter_kept_fraction"] for k in ks), max(dg[k]["relu_filter_kept_fraction"] for k in ks), " ".join("%.0e" % v for v in d), max(d)))
