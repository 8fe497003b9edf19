"""CPU suite: the HIP kernel sources executed by the wave emulator (tests/emu) against the oracle / reference goldens.

This checks kernel index algebra, weight packing, LDS addressing and barrier placement without a GPU.  It says nothing
about the hardware; `tests/test_gpu_parity.py` (-m gpu) runs the same cases through the product library on an MI355X.
"""
import parity_cases as P


def test_rays(emu):
    P.case_rays(emu)


def test_posenc(emu):
    P.case_posenc(emu)


def test_stratified(emu):
    P.case_stratified(emu)


def test_cumprod(emu):
    P.case_cumprod(emu)


def test_volume_render(emu):
    P.case_volume_render(emu)


def test_volume_render_bwd(emu):
    P.case_volume_render_bwd(emu)


def test_sample_pdf(emu):
    P.case_sample_pdf(emu)


def test_loss_adam(emu):
    P.case_loss_adam(emu)


def test_mlp_forward(emu):
    P.case_mlp_forward(emu, names=("default4x128", "fern8x128_skip3_L6", "novw4x128", "noinput_linear"), m=40)


def test_mlp_forward_northstar(emu):
    P.case_mlp_forward(emu, names=("northstar8x256",), m=33)


def test_mlp_extreme_geometries(emu):
    """num_layers 1 and 16, a skip connection at every layer, no view directions with a 256-wide net."""
    names = ("one_layer", "one_layer_novw_256", "sixteen_layers_skip5", "skip_every_layer_256")
    P.case_mlp_forward(emu, names=names, m=37)
    P.case_mlp_backward(emu, names=names, m=45)


def test_mlp_many_layers_and_jobs(emu):
    """16 layers with a skip connection at every layer = 35 weight-gradient jobs (the device job table once held 32), and
    more than 16 layers (nerf/models.py:186-196 takes any num_layers)."""
    names = ("sixteen_layers_skip1", "twenty_layers_skip7")
    P.case_mlp_forward(emu, names=names, m=33)
    P.case_mlp_backward(emu, names=names, m=40)


def test_mlp_64_wide_instances(emu):
    """hidden_size 64 (config/llff.yml:49, pretrained/fern-lowres/config.yml:19): its own kernel instances, not the
    128-wide ones with 4x the FLOPs."""
    names = ("llff4x64_skip3_L6", "deep8x64_skip4", "novw3x64_skip1", "one_layer_64")
    P.case_mlp_forward(emu, names=names, m=70)
    P.case_mlp_backward(emu, names=names, m=150)
    P.case_mlp_input_grad(emu, names=("llff4x64_skip3_L6",), m=45)


def test_render_64_wide_vs_oracle(emu):
    P.case_render_vs_oracle(emu, P.MLP_GEOMETRIES["llff4x64_skip3_L6"], n=12, nc=16, nf=16, with_grads=True, tag="llff64",
                            grad_tol=(1e-3, 2e-2))


def test_mlp_512_wide_instances(emu):
    """hidden_size above 256 (nerf/models.py:186-196 takes any): 512-wide kernel instances, 512-row stash regions read
    by the weight-gradient kernel one 256-row half at a time, four 32-bit words of ReLU bits per lane."""
    names = ("wide3x512_skip2", "wide2x320", "novw2x512")
    P.case_mlp_forward(emu, names=names, m=37)
    P.case_mlp_backward(emu, names=names, m=70)
    P.case_mlp_input_grad(emu, names=("wide3x512_skip2", "wide2x320"), m=45)


def test_mlp_forward_f16x3(emu):
    """The inference forward on fp16 pieces: ~3 x 2^-24 per product -- the fp32 kernels' 2e-5 bound and an fp32-sized distance
    from the fp64 forward, every layer kind, both widths, the persistent loop."""
    P.case_mlp_forward_f16x3(emu, m=37, precision=P.F16X3)
    P.case_mlp_forward_f16x3(emu, names=("default4x128",), m=900, precision=P.F16X3)
    P.case_render_f16x3(emu, P.MLP_GEOMETRIES["default4x128"], n=10, nc=8, nf=8, tag="4x128_emu", precision=P.F16X3)


def test_mlp_f16x3_training_levels_hold_the_fp32_gradient_bounds(emu):
    """F16X3_FWD / _FWD_DGRAD / _TRAIN: parameter and input gradients within the fp32 kernels' 2e-5 of max|g| (same ReLU margin,
    same row counts), the fused render with gradients at the fp32 tolerances; the data-gradient chain runs on d(raw output) times
    a power of two taken from its maximum (tiny cotangents: the fp16 pieces would otherwise flush them)."""
    P.case_mlp_backward(emu, names=("default4x128", "skip_every_layer_256"), m=120, precision=P.F16X3_FWD)
    P.case_mlp_backward(emu, names=("fern8x128_skip3_L6", "novw4x128", "one_layer"), m=120, precision=P.F16X3_FWD_DGRAD)
    P.case_mlp_backward(emu, names=("skip_every_layer_256", "one_layer_novw_256", "default4x128"), m=120, precision=P.F16X3_TRAIN)
    P.case_mlp_backward(emu, names=("default4x128",), m=100, precision=P.F16X3_TRAIN, g_scale=3e-7)
    # weights 4x torch's init: activations in the thousands, d(pre-activation) a million times d(raw output) -- no fixed fp16 scale
    # holds both ends; the per-sample exponents (forward and data gradient) and the per-region scales of k_wgrad_f16x3 do
    P.case_mlp_backward(emu, names=("skip_every_layer_256",), m=100, precision=P.F16X3_TRAIN, w_gain=4.0, g_scale=1e-4)
    P.case_mlp_backward(emu, names=("default4x128",), m=100, precision=P.F16X3_FWD_DGRAD, w_gain=0.25)
    P.case_mlp_input_grad(emu, names=("default4x128", "novw4x128"), m=45, precision=P.F16X3_FWD_DGRAD)
    P.case_render_vs_oracle(emu, P.MLP_GEOMETRIES["default4x128"], n=12, nc=16, nf=16, with_grads=True, tag="f16x3_train_emu",
                            precision=P.F16X3_TRAIN)  # (128-wide nets: the four full blocks reach k_wgrad_f16x3<128, 128>)


def test_compacted_backward_equals_dense(emu):
    """nerfhip_plan_set_bwd_compaction: the backward over the samples whose d(raw output) row is not all zero == the dense backward,
    every kernel family (fp32 narrow / wide / 64- and 512-wide; fp16-piece data gradient and weight gradient), zero fractions 0 ... 1."""
    P.case_mlp_backward_compacted(emu, names=("default4x128", "novw4x128"), m=200, fractions=(0.0, 0.45, 1.0))
    P.case_mlp_backward_compacted(emu, names=("skip_every_layer_256", "llff4x64_skip3_L6"), m=140, fractions=(0.6,))
    P.case_mlp_backward_compacted(emu, names=("wide3x512_skip2",), m=140, fractions=(0.5,))
    P.case_mlp_backward_compacted(emu, names=("one_layer",), m=140, precision=P.F16X3_FWD_DGRAD, fractions=(0.7,))
    P.case_mlp_backward_compacted(emu, names=("skip_every_layer_256",), m=140, precision=P.F16X3_TRAIN, fractions=(0.5,))
    P.case_mlp_backward_compacted(emu, names=("default4x128",), m=160, precision=P.F16X3_TRAIN, fractions=(0.0, 0.6), g_scale=3e-7)


def test_render_backward_modes_dense_compacted_recomputed(emu):
    """The fused render with both plans dense / compacted / compacted + recomputed: identical outputs, equal gradients (fp32 and fp16
    pieces; 128- and 64-wide kernels on the emulator, 256-wide on the GPU)."""
    P.case_render_compacted(emu, P.MLP_GEOMETRIES["default4x128"], n=10, nc=16, nf=16, tag="4x128_emu")
    P.case_render_compacted(emu, P.MLP_GEOMETRIES["llff4x64_skip3_L6"], n=9, nc=16, nf=16, tag="llff64_emu", white=True)
    P.case_render_compacted(emu, P.MLP_GEOMETRIES["default4x128"], n=10, nc=16, nf=16, precision=P.F16X3_TRAIN, tag="4x128_emu")
    P.case_render_compacted(emu, P.MLP_GEOMETRIES["novw4x128"], n=9, nc=16, nf=8, precision=P.F16X3_FWD_DGRAD, tag="novw_emu")
    # ... and d(loss)/d(rays) through the compacted images (nerfhip_render_bwd_rays), both modes
    P.case_ray_grad(emu, P.MLP_GEOMETRIES["default4x128"], n=12, nc=8, nf=8, compact=True)
    P.case_ray_grad(emu, P.MLP_GEOMETRIES["default4x128"], n=12, nc=8, nf=8, compact="recompute")


def test_fused_backward_of_64_wide_nets(emu):
    """nerfhip_plan_set_bwd_compaction(plan, 3 / 4), csrc/mlp64r.hip: persistent workgroups with the whole net in LDS -- forward
    recomputed, data gradient, weight gradient and bias sums in one kernel, a fixed-order reduction of one partial per workgroup --
    against the dense three-kernel backward on the same batch: 4 layers (config/fern.yml's 4 x 64), one layer, a padded hidden size
    (40 of 64 units); several rounds per workgroup (the emulator has 3 "compute units"), a ragged last round; over every sample and
    over the compaction list.  Plans without a resident image refuse the modes."""
    import pytest
    P.case_render_compacted(emu, P.MLP_GEOMETRIES["llff4x64_skip3_L6"], n=9, nc=16, nf=16, tag="llff64_fused_emu", white=True, fused=True)
    P.case_render_compacted(emu, P.MLP_GEOMETRIES["llff4x64_skip3_L6"], n=37, nc=24, nf=8, tag="llff64_fused_rounds_emu", noise=1.0, fused=True)
    P.case_render_compacted(emu, P.MLP_GEOMETRIES["one_layer_64"], n=20, nc=16, nf=16, tag="one64_fused_emu", noise=0.0, fused=True)
    P.case_render_compacted(emu, P.MLP_GEOMETRIES["narrow3x40"], n=17, nc=24, nf=16, tag="narrow40_fused_emu", fused=True)
    P.case_render_compacted(emu, P.MLP_GEOMETRIES["two_layer_64"], n=11, nc=16, nf=16, tag="two64_fused_emu", fused=True)
    P.case_render_compacted(emu, P.MLP_GEOMETRIES["three_layer_48"], n=11, nc=16, nf=8, tag="three48_fused_emu", white=True, fused=True)
    P.case_render_fused_edges(emu, P.MLP_GEOMETRIES["llff4x64_skip3_L6"])
    # ... and d(loss)/d(rays) with the fused modes set: the ray gradient needs the d(pre-activation) images -> mode 2's data flow
    P.case_ray_grad(emu, P.MLP_GEOMETRIES["llff4x64_skip3_L6"], n=12, nc=8, nf=8, compact="fused_compact")
    # (mode 5: the forward left the register-image stash; the recomputing flow overwrites it with the general one for the list)
    P.case_ray_grad(emu, P.MLP_GEOMETRIES["llff4x64_skip3_L6"], n=12, nc=8, nf=8, compact="fused_stash")
    for name in ("default4x128", "novw3x64_skip1", "deep8x64_skip4"):   # 128 wide / no view directions / 8 layers with a skip layer
        plan = emu.make_plan(P.MLP_GEOMETRIES[name], 0)
        with pytest.raises(Exception, match="fused backward"):
            emu.set_compaction(plan, "fused")
        emu.lib.plan_destroy(plan)


def test_f16x3_scale_fuzz(emu):
    """(the corners on the emulator; the GPU suite walks the whole 3 x 3 x 3 grid on four geometries)"""
    P.case_f16x3_scale_fuzz(emu, m=24, names=("default4x128", "deep8x128_skip4"),
                            grid=[(1.0, 1e-3, 0.0), (1e-6, 1e-3, 0.0), (1e3, 30.0, 100.0), (1e-6, 30.0, 1.0), (1.0, 3e-2, 0.0), (1.0, 1.0, 1.0)])


def test_f16x3_dead_layers(emu):
    P.case_f16x3_dead_layers(emu, m=60)


def test_f16x3_range_extremes(emu):
    P.case_f16x3_range_extremes(emu, m=60)


def test_mlp_f16x3_64_wide_instances(emu):
    """k_mlp_fwd_f16x3w<64> / k_mlp_dgrad_f16x3w<64>: four output tiles, two 32-deep k-blocks, whole layers inside one chunk."""
    names = ("llff4x64_skip3_L6", "novw3x64_skip1", "one_layer_64")
    P.case_mlp_forward_f16x3(emu, names=names + ("narrow3x40",), m=37, precision=P.F16X3)
    P.case_mlp_backward(emu, names=names, m=100, precision=P.F16X3_TRAIN)
    P.case_mlp_backward(emu, names=("deep8x64_skip4",), m=100, precision=P.F16X3_FWD)
    P.case_render_vs_oracle(emu, P.MLP_GEOMETRIES["llff4x64_skip3_L6"], n=12, nc=16, nf=16, noise=1.0, with_grads=True,
                            tag="f16x3_llff64_emu", grad_tol=(1e-3, 2e-2), precision=P.F16X3_TRAIN)


def test_ndc_rays_backward(emu):
    P.case_ndc_rays_bwd(emu, n=200)


def test_mlp_extended_encodings(emu):
    """num_encoding_fn_xyz up to 16, num_encoding_fn_dir up to 10 (nerf/models.py:198-201 takes any): the extended slot
    registers of the forward kernel, every width."""
    P.case_mlp_forward(emu, names=P.EXT_GEOMETRIES, m=37)
    P.case_mlp_backward(emu, names=P.EXT_GEOMETRIES, m=45)
    P.case_mlp_input_grad(emu, names=("L12_4x128", "Ld5_4x128_skip2"), m=45)
    P.case_render_vs_oracle(emu, P.MLP_GEOMETRIES["L12_4x128"], n=10, nc=8, nf=8, with_grads=True, tag="L12_emu")
    P.case_ray_grad(emu, P.MLP_GEOMETRIES["L12_4x128"], n=8)


def test_mlp_padded_hidden_sizes(emu):
    """hidden_size other than 128 / 256 (the reference constructor takes any: nerf/models.py:185-196), odd included."""
    names = ("narrow3x40", "odd5x99_skip2", "wide3x200_skip1", "novw2x130")
    P.case_mlp_forward(emu, names=names, m=37)
    P.case_mlp_backward(emu, names=names, m=45)


def test_mlp_input_gradient(emu):
    P.case_mlp_input_grad(emu, m=45)


def test_mlp_golden(emu):
    P.case_mlp_golden(emu)


def test_mlp_backward(emu):
    P.case_mlp_backward(emu, names=("default4x128", "fern8x128_skip3_L6", "novw4x128"), m=70)


def test_e2e_golden_a(emu):
    P.case_e2e_golden(emu, "e2e_a.npz")


def test_e2e_golden_d_noviewdirs(emu):
    P.case_e2e_golden(emu, "e2e_d.npz")


def test_internal_rng(emu):
    P.case_internal_rng(emu)


def test_edge_cases(emu):
    P.case_edges(emu)


def test_select_rays(emu):
    P.case_select(emu)


def test_select_uniformity(emu):
    P.case_select_uniformity(emu)


def test_image_output(emu):
    P.case_image_output(emu)


def test_e2e_northstar_reference_golden(emu):
    P.case_e2e_northstar_golden(emu)


def test_ray_gradients(emu):
    P.case_ray_grad(emu, P.MLP_GEOMETRIES["default4x128"])
    P.case_ray_grad(emu, P.MLP_GEOMETRIES["novw3x64_skip1"], n=10, white=True, noise=0.5)
