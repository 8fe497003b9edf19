"""Every numeric bound of the full-batch GPU suite (tests/test_gpu_fullsize.py) and the per-arithmetic bounds of tests/test_gpu_parity.py,
in ONE table, each with where it comes from.

Rule of the table: a bound belongs to the QUANTITY, not to the arithmetic that computes it.  The
fp32-grade arithmetics -- "fp32" (v_mfma_f32_16x16x4_f32 / 32x32x2: the reference's own arithmetic) and the fp16-piece plans "f16x3",
"f16x3_train" (three fp16 MFMAs per product, ~3 x 2^-24 per product) -- are held to the SAME numbers: `bound(name, arith)` has no
per-arithmetic entry for them, `OVERRIDES` must stay empty for them, and tests/test_host_abi.py::test_tolerance_table_* asserts both
(plus that the test sources select no bound by arithmetic).  Round 4 had loosened two bounds for every arithmetic and branched two more
on `arith != "fp32"` when the two-wave fp16 kernels met them; what replaced those four:

  * gradients teacher-forced on the oracle's depths: compared on the samples whose ReLU branches round-off cannot decide
    (`RELU_MARGIN`, the filter tests/parity_cases.py::case_mlp_backward always had), and `max <= 1e-4` again;
  * 256-ray slice against fp64: same filter, and the ONE assertion the fp32 kernels had, for everybody;
  * coarse-net gradients of the fern batch: the fp64 yardstick for everybody (no further from the exact gradient than torch's own fp32
    gradient is, x 1.5) instead of 5 x a distance measured between two fp32 evaluations that share their rounding;
  * coarse maps of the trained lego nets at eval size: the two yardsticks the fine maps always had -- the reference's own
    torch-on-cuda vs torch-on-CPU spread and the distance of its fp32 run from an fp64 run -- instead of one number (2e-5) set
    with 20 % headroom from the first implementation measured (on this checkpoint fp32 itself is 1.4e-4 from fp64).

A value is either a number, a (max, p99.9) pair, or a dict of the yardstick-relative form `got <= mul * yardstick + add`.
"""

TOL = {
    # ---- the ReLU-margin filter (oracle/nerf_oracle.py::mlp_relu_margin) -------------------------------------------------------
    "relu_margin": (1e-5, "a sample is compared on gradients only if no ReLU input of it is within 1e-5 (relative to the sample's "
                          "largest pre-activation) of zero: two fp32-grade evaluations differ by ~sqrt(256) * 2^-24 ~ 1e-6 of the scale per "
                          "layer, a few 1e-6 over eight layers; 10x the unit test's 1e-6 (m = 1500) because a full batch holds 1e9 "
                          "pre-activations"),
    "relu_margin.min_kept": (0.75, "at least this fraction of the samples survives the filter (8x256: 2,304 ReLU inputs per sample), or "
                                   "the comparison tests nothing"),
    # ---- end to end, full batches (tests/test_gpu_fullsize.py::_end_to_end_on) --------------------------------------------------
    "e2e.coarse_maps.max": (1e-5, "rgb / acc / depth of the coarse pass, no sampler in front: fp32 round-off (measured 1-3e-6)"),
    "e2e.acc_fine.max": (5e-4, "behind the sampler (SURVEY 0.11: the reference itself moves by 3.6e-4 between fp32 and fp64)"),
    "e2e.depth_fine.max": (2e-3, "behind the sampler; ray for ray the reference's own cuda-vs-CPU spread (DESIGN 3)"),
    "e2e.rgb_fine": ((1e-4, 1e-4), "the north star's bar on colour: max and p99.9 (BASELINE.json)"),
    "e2e.rgb_fine.padded5x99": ((3e-4, 1e-4), "not a BASELINE configuration: ONE ray of 2048 lands a fine sample in a neighbouring bin "
                                              "(measured max 1.6e-4, p99.9 7.6e-5)"),
    "e2e.yardstick.rays_over_1e4": (dict(mul=2.0, add=3), "no more rays beyond 1e-4 than the reference's own torch-on-cuda vs torch-on-CPU "
                                                          "pair has, x 2 + 3 (both counts are a handful of chaotic events)"),
    "e2e.yardstick.p999": (dict(mul=2.0, add=2e-6), "bulk no wider than that pair's"),
    "e2e.yardstick.rays_moved": (dict(mul=2.0, add=3), "rays with a fine sample in another bin than the oracle's"),
    "e2e.loss": (1e-5, "the summed coarse + fine mse"),
    "e2e.grad_coarse.lego8x256": ((2e-4, 5e-5), "of max|g| per tensor: two fp32-grade sums of 262,144 terms (measured 1.1-1.4e-4 / 3.6e-5)"),
    "e2e.grad_coarse.default4x128": ((1e-4, 5e-5), "measured 2.0e-5 / 1.0e-5"),
    "e2e.grad_coarse.padded5x99": ((1.7e-4, 5e-5), "measured 3.4e-5 / 1e-5"),
    "e2e.grad_coarse.fp64_yardstick": (dict(mul=1.5, add=1e-6, cap=1e-4),
                                       "the fern batches, 8x128 and 4x64 (sigma noise 1.0: the early layers' cotangents nearly cancel): no further from the fp64 "
                                       "gradient than torch's own fp32 gradient is, x 1.5, and inside the lego batches' 1e-4 of the fp32 oracle"),
    "e2e.grad_fine.lego8x256": ((3e-3, 1e-3), "behind the sampler: measured 6.5e-4 / 3.6e-4"),
    "e2e.grad_fine.default4x128": ((1.3e-3, 9e-4), "measured 2.6e-4 / 1.9e-4"),
    "e2e.grad_fine.padded5x99": ((3e-3, 2e-3), "measured 6e-4 / 4e-4"),
    "e2e.grad_fine.yardstick": (dict(mul=2.0, add=2e-5),
                                "the fern batches, 8x128 and 4x64, fine net (behind the sampler): no further from the CPU oracle's gradient than the "
                                "reference's OWN torch-on-cuda gradient of the same batch is, x 2 (each side is a handful of moved fine samples / "
                                "flipped ReLU branches), + the fp32 floor of a teacher-forced gradient (tf.grad.filtered measures 2.3e-5).  Round 5 "
                                "held these to 5 x the first measurement (3.2e-5 / 5.1e-5): a trip-wire, not a bound"),
    # ---- teacher-forced fine pass (the oracle's depths; _teacher_forced_on) ----------------------------------------------------
    "tf.raw.max": (1e-6, "raw network outputs on identical inputs (measured 9e-8)"),
    "tf.maps.max": (2e-6, "rgb / acc composited from them (measured 4e-7)"),
    "tf.depth.max_over_far": (2e-6, "depth = sum w z on the same depths, relative to `far`"),
    "tf.disp.rel": (1e-5, "disparity, relative"),
    "tf.grad.filtered": ((1e-4, 5e-5), "every parameter gradient of the fine net over the samples that pass `relu_margin`, of max|g| per "
                                       "tensor: max and p99.9.  The bound round 3 set for the unfiltered batch; what exceeded it in round 4 "
                                       "(1.12e-4, one row) was one flipped ReLU branch of one sample"),
    "tf.slice.vs_fp64.p999": (dict(mul=1.5, add=1e-6), "256 filtered rays: no further from an fp64 run of the oracle than torch's fp32 run is"),
    "tf.slice.vs_fp64.max": (dict(mul=1.5, add=1e-4), "the same for the largest entry"),
    # ---- BASELINE configs[0]: tiny_nerf.py through the helpers (the user's network runs on torch's own GEMMs on both sides) --------
    "tiny.rgb": (2e-5, "rgb map of a 3-layer MLP composited through this package's helpers vs the oracle's: fp32 round-off (rocBLAS vs MKL)"),
    "tiny.loss": (1e-6, "the mse of that rgb map against the target"),
    "tiny.grad": ((2e-4, 1e-3), "gradients into the user's torch model: atol of max|g|, rtol"),
    # ---- sampler ------------------------------------------------------------------------------------------------------------------
    "sampler.index_flips": (5, "inverse-CDF indices that differ from torch's, of 524,288 (measured 0-1; SURVEY H3: 3.8e-6 per ulp)"),
    # ---- eval size (config 5), inference instantiation ---------------------------------------------------------------------------
    "eval.coarse_maps.max": (1e-5, "synthetic scenes, no sampler in front"),
    "eval.rgb_fine.p999": (dict(mul=2.0, floor=1e-4), "the north star's bar for the bulk unless the reference's own pair is wider"),
    "eval.yardstick.rays_over_1e4": (dict(mul=2.0, add=3), "as e2e.yardstick.rays_over_1e4, at eval size"),
    "eval.yardstick.p999": (dict(mul=2.0, add=2e-6), "as e2e.yardstick.p999, at eval size"),
    "eval.smooth.rgb_fine.p999": (1.5e-4, "8x256 scene conditioned like a trained one (measured 1.07e-4; torch-on-cuda 8.7e-5)"),
    "eval.smooth.rays_over_1e4.fraction": (0.0025, "measured 25 of 16,384 (torch-on-cuda 21)"),
    "eval.trained.coarse_maps.yardstick": (dict(mul=1.5, add=2e-6),
                                           "rgb / acc of the coarse pass of the TRAINED lego nets (raw outputs up to 1e4, sigma 4e3: ulp-level "
                                           "differences of the device's sin / cos and summation order come out at ~2e-5 of acc): no wider than "
                                           "the reference's own torch-on-cuda vs torch-on-CPU pair, x 1.5.  Was a flat 2e-5 set 20 % above the first "
                                           "implementation measured (fp32 kernels 1.63e-5, torch-on-cuda 1.73e-5, fp16 pieces 1.67e-5 / 2.24e-5)"),
    "eval.trained.coarse_maps.fp64_yardstick": (dict(mul=1.5, add=2e-6),
                                                "... and no further from an fp64 run of the oracle than the oracle's fp32 run is (1.4e-4 of acc on "
                                                "this checkpoint: what fp32 itself loses here, SURVEY 0.11)"),
    "eval.trained.yardstick.rays_over_1e4": (dict(mul=2.0, add=8), "trained nets: the reference moves by 6e-4 between fp32 and fp64"),
    "eval.trained.yardstick.p999": (dict(mul=2.0, add=5e-6), "bulk no wider than that pair's, with the slack of a trained scene"),
    "eval.trained.rgb_fine.p999": (2e-4, "SURVEY 0.11"),
    # ---- per-arithmetic unit cases of tests/test_gpu_parity.py (tests/parity_cases.py) ------------------------------------------
    "unit.mlp_fwd": ((2e-5, 2e-5), "MLP forward vs the oracle: atol, rtol (case_mlp_forward)"),
    "unit.mlp_fwd.vs_fp64_over_scale": (1.5e-6, "distance from the fp64 forward over the output scale (torch fp32: 1.2-3.6e-7)"),
    "unit.mlp_bwd": (dict(margin=1e-6, tol=2e-5), "teacher-forced MLP backward, m = 1500: of max|g| per tensor (case_mlp_backward)"),
    "unit.mlp_input_grad": (2e-5, "d(loss)/d(x), of max|g|"),
    "e2e.compact_vs_dense": (2e-5, "the full lego batch's backward compacted against dense, of max|g| per tensor: two fp32-grade orders of one "
                                   "sum of up to 786,432 terms whose zero terms were dropped (measured on MI355X: see "
                                   "profiles/r06_parity_small_cases.json `full_size_compact_vs_dense_*`)"),
    "unit.compact_vs_dense": (1e-5, "the compacted backward against the dense one of the same plan, of max|g| per tensor: the same terms "
                                    "(zero terms dropped) summed under another split of the sample range over the workgroups -- two "
                                    "fp32-grade orders of one sum, ~sqrt(terms) x 2^-24 of a term (case_mlp_backward_compacted; measured "
                                    "<= 1e-6 on the emulator and on MI355X)"),
    "unit.render_grad.coarse_fp64_yardstick": (dict(mul=1.5, add=1e-6),
                                               "coarse-net gradients of the small fused-render cases (48 rays of 8x256; 200 rays of 4x128 with a white "
                                               "background and sigma noise 1.0, whose early-layer cotangents nearly cancel): no further from an fp64 run of "
                                               "the oracle than the oracle's own fp32 run is, x 1.5, per tensor of max|g|.  Round 5 held them to 5 x the first "
                                               "measurement"),
    "unit.render_grad.fine_sanity": (1e-2, "fine-net gradients of the same small cases END TO END: behind the sampler a handful of rays decide the "
                                           "maximum (one moved fine sample flips ReLU branches: any two fp32 evaluations are ~1e-3 apart on 48-200 rays) -- "
                                           "a sanity cap, NOT the parity claim: that is the teacher-forced `tf.grad.filtered` (1e-4) at full size and "
                                           "`unit.mlp_bwd` (2e-5) on the kernels, and `e2e.grad_fine.yardstick` where the batch is large enough for statistics"),
}

# arithmetic -> {name: value}.  MUST hold no entry for the fp32-grade arithmetics (asserted on the CPU).
OVERRIDES = {}
FP32_GRADE = ("fp32", "f16x3", "f16x3_train", "f16x3_fwd", "f16x3_fwd_dgrad")


def bound(name, arith="fp32"):
    v = OVERRIDES.get(arith, {}).get(name)
    return TOL[name][0] if v is None else v


def provenance(name):
    return TOL[name][1]


def within(got, yardstick, form):
    """`got <= mul * yardstick + add` (optionally: at least `floor`, at most `cap`)."""
    lim = form.get("mul", 1.0) * yardstick + form.get("add", 0.0)
    if "floor" in form:
        lim = max(lim, form["floor"])
    return got <= lim
