"""Drop-in for ``nerf/train_utils.py``: run_network, predict_and_render_radiance, run_one_iter_of_nerf with the
reference's signatures, defaults and return layout (nerf/train_utils.py:8-202).

When both networks are this package's FlexibleNeRFModel and the encoders come from this package's
get_embedding_function, predict_and_render_radiance runs the fused pipeline of libnerfhip.so (one C-ABI call forward,
one backward; encodings and activations never reach HBM).  Otherwise it composes the unit kernels exactly like the
reference composes its torch ops, so arbitrary user networks still work.

Random draws: made with torch.rand / torch.randn on the rays' device in the reference's order and shapes
(t_rand, coarse noise, u, fine noise per ray chunk), then handed to the kernels -- seeding torch reproduces a run.

Reference quirk kept on purpose: run_one_iter_of_nerf does not forward `mode` to predict_and_render_radiance
(train_utils.py:171-181), so rendering always reads options.nerf.train.* ; only the ray chunk size and the output
reshape honour `mode`.
"""
import ctypes as C

import torch

from . import _lib as L
from .models import FlexibleNeRFModel
from .nerf_helpers import EmbeddingFunction, get_minibatches, linspace01, ndc_rays, sample_pdf_2 as sample_pdf
from .volume_rendering_utils import volume_render_radiance_field


def run_network(network_fn, pts, ray_batch, chunksize, embed_fn, embeddirs_fn):
    """nerf/train_utils.py:8-25."""
    pts_flat = pts.reshape((-1, pts.shape[-1]))
    embedded = embed_fn(pts_flat)
    if embeddirs_fn is not None:
        viewdirs = ray_batch[..., None, -3:]
        input_dirs = viewdirs.expand(pts.shape)
        input_dirs_flat = input_dirs.reshape((-1, input_dirs.shape[-1]))
        embedded_dirs = embeddirs_fn(input_dirs_flat)
        embedded = torch.cat((embedded, embedded_dirs), dim=-1)
    batches = get_minibatches(embedded, chunksize=chunksize)
    preds = [network_fn(batch) for batch in batches]
    radiance_field = torch.cat(preds, dim=0)
    return radiance_field.reshape(list(pts.shape[:-1]) + [radiance_field.shape[-1]])


# ---- fused path -------------------------------------------------------------------------------------------------------
def _fusable(model_coarse, model_fine, enc_xyz, enc_dir, num_fine):
    if not isinstance(model_coarse, FlexibleNeRFModel):
        return False
    if num_fine > 0 and not isinstance(model_fine, FlexibleNeRFModel):
        return False
    models = [model_coarse] + ([model_fine] if num_fine > 0 else [])
    for m in models:
        c = m.cfg
        if not isinstance(enc_xyz, EmbeddingFunction):
            return False
        if (enc_xyz.num_encoding_functions, bool(enc_xyz.include_input), bool(enc_xyz.log_sampling)) != (
                c["num_encoding_fn_xyz"], c["include_input_xyz"], c["log_sampling_xyz"]):
            return False
        if c["use_viewdirs"]:
            if not isinstance(enc_dir, EmbeddingFunction):
                return False
            if (enc_dir.num_encoding_functions, bool(enc_dir.include_input), bool(enc_dir.log_sampling)) != (
                    c["num_encoding_fn_dir"], c["include_input_dir"], c["log_sampling_dir"]):
                return False
        elif enc_dir is not None:
            return False
    return True


class _FusedRender(torch.autograd.Function):
    """The fused pipeline as one autograd node.  Differentiable outputs: the colour, accumulation and depth maps of both
    passes (what any loss built on the reference's outputs can touch; disparity is derived from depth and acc by the
    caller in torch, so it is covered too).  Inputs with a gradient: the parameters of the two nets, passed one by one
    (they alias each model's flat buffer; backward returns each its slice of the flat gradient).

    With `rays_grad` the ray batch itself receives a gradient (pose optimisation: under autograd the reference
    differentiates pts = ro + rd * z, train_utils.py:67,107, and dists * ||rd||, volume_rendering_utils.py:24): columns
    0..5 (origin, direction) and 8..10 (viewdirs) of d(loss)/d(rays); near / far are constants of the ray.

    The workspace (activation stash, ~4.8 MB per ray for the 8x256 nets) is allocated per call from torch's caching
    allocator and owned by the autograd node: several nodes may be alive at once (run_one_iter_of_nerf renders a batch
    in ray chunks and backpropagates afterwards), so it must not be shared between calls."""

    @staticmethod
    def forward(ctx, rays, model_c, model_f, cfg_tuple, rand, training, rays_grad, *params):
        lib = L.get_lib()
        nc, nf, perturb, lindisp, white, noise_std = cfg_tuple
        n, stride = rays.shape
        dev = rays.device
        cfg = L.RenderCfg(nc, nf, int(bool(perturb)), int(bool(lindisp)), int(bool(white)), float(noise_std), stride)
        # `training` (decided by the caller: grad mode is off inside Function.forward, and needs_input_grad ignores
        # torch.no_grad()): keep the activation stash for a backward
        # (inference -- no backward follows -- runs on each model's inference plan: fp32 unless set_inference_precision
        # chose the fp16-piece kernels)
        if training:  # (models with set_backward_compaction("auto") pick the mode of this pass from their last compacted backward)
            model_c._auto_choose_backward()
            if nf > 0:
                model_f._auto_choose_backward()
        plan_c = model_c._plan if training else model_c._inference_plan()
        plan_f = (model_f._plan if training else model_f._inference_plan()) if nf > 0 else None
        # (training layout 2: this node's backward runs the two nets one after the other on one stream, so they share one
        # set of backward buffers -- several nodes may be alive at once when a batch is rendered in ray chunks)
        wsb = lib.render_workspace_bytes(plan_c, plan_f, C.byref(cfg), n, 2 if training else 0)
        if wsb < 0:
            raise L.NerfHipError(lib.last_error().decode())
        ws = torch.empty(wsb // 4 + 1, dtype=torch.float32, device=dev)
        ctx.set_materialize_grads(False)  # cotangents of outputs the loss never touched arrive as None, not as zeros
        names = ("rgb_coarse", "disp_coarse", "acc_coarse", "depth_coarse", "rgb_fine", "disp_fine", "acc_fine",
                 "depth_fine")
        bufs = {k: torch.empty((n, 3) if k.startswith("rgb") else (n,), dtype=torch.float32, device=dev) for k in names}
        out = L.RenderOut(*[bufs[k].data_ptr() for k in names])
        rr = L.RenderRand(*[None if r is None else r.data_ptr() for r in rand])
        packed_c = model_c._packed() if training else model_c._inference_packed()
        packed_f = (model_f._packed() if training else model_f._inference_packed()) if nf > 0 else None
        with L.launch_on(rays, ws, packed_c, packed_f, *[r for r in rand if r is not None]) as st:
            lib.render_fwd(plan_c, plan_f, C.byref(cfg), rays.data_ptr(), n, packed_c.data_ptr(),
                           packed_f.data_ptr() if packed_f is not None else None, linspace01(nc, dev).data_ptr(),
                           linspace01(nf, dev).data_ptr() if nf > 0 else None, C.byref(rr), 0, 0, C.byref(out),
                           ws.data_ptr(), wsb, 2 if training else 0, st)  # (2: this node's backward shares one set of backward buffers)
        # (the ray gradient multiplies by the weights of THIS forward: keep copies only if it will be asked for)
        flats = (model_c._flat.clone(), model_f._flat.clone() if nf > 0 else None) if (training and rays_grad) else None
        ctx.keep = (rays, model_c, model_f, cfg, rand, ws, wsb, packed_c, packed_f, training, flats)
        # (the backward data flow is an option of the PLAN, and what this forward left in the workspace -- the general stash, nothing, the
        # register-image stash -- depends on it: the node's backward runs in the mode its forward ran in, whatever
        # set_backward_compaction was called with in between)
        ctx.bwd_modes = (lib.plan_bwd_compaction(plan_c), lib.plan_bwd_compaction(plan_f) if plan_f is not None else 0) if training else None
        ctx.mark_non_differentiable(bufs["disp_coarse"], bufs["disp_fine"])
        return tuple(bufs[k] for k in names)

    @staticmethod
    def backward(ctx, g_rgb_c, g_disp_c, g_acc_c, g_depth_c, g_rgb_f, g_disp_f, g_acc_f, g_depth_f):
        lib = L.get_lib()
        rays, model_c, model_f, cfg, rand, ws, wsb, packed_c, packed_f, training, flats = ctx.keep
        if not training:
            raise RuntimeError("fused render was run without gradient bookkeeping")
        n = rays.shape[0]
        dev = rays.device
        nf = cfg.num_fine
        keep = [None if g is None else g.contiguous().float()
                for g in (g_rgb_c, g_acc_c, g_depth_c, g_rgb_f, g_acc_f, g_depth_f)]
        parts = 0
        if any(k is not None for k in keep[:3]):
            parts |= L.PART_COARSE
        if nf > 0 and any(k is not None for k in keep[3:]):
            parts |= L.PART_FINE
        gpc = torch.zeros(model_c.num_flat_params, dtype=torch.float32, device=dev)
        gpf = torch.zeros(model_f.num_flat_params, dtype=torch.float32, device=dev) if nf > 0 else None
        g_rays = torch.zeros_like(rays) if flats is not None else None
        if parts:
            cot = L.RenderCotangents(*[None if k is None else k.data_ptr() for k in keep])
            rr = L.RenderRand(*[None if r is None else r.data_ptr() for r in rand])
            plan_f = model_f._plan if nf > 0 else None
            tmp, tmpb = None, 0
            if g_rays is not None:
                tmpb = lib.render_bwd_rays_tmp_bytes(model_c._plan, plan_f, C.byref(cfg), n)
                tmp = torch.empty(tmpb // 4 + 1, dtype=torch.float32, device=dev)
            now = (lib.plan_bwd_compaction(model_c._plan), lib.plan_bwd_compaction(plan_f) if plan_f is not None else 0)
            for plan, was, cur in ((model_c._plan, ctx.bwd_modes[0], now[0]), (plan_f, ctx.bwd_modes[1], now[1])):
                if plan is not None and was != cur:
                    lib.plan_set_bwd_compaction(plan, was)
            # "auto" models read the {kept, total} words of their compacted backward; the two nets share one set of backward buffers
            # here (the fine pass runs first), so the passes are issued one by one with the copy in between -- the same launches in
            # the same order as the single call (not with a ray gradient: its second pass accumulates into the first one's)
            auto = [m for m in (model_c, model_f) if m is not None and getattr(m, "_backward_choice", None) == "auto"]
            passes = [parts]
            if auto and g_rays is None and parts == (L.PART_COARSE | L.PART_FINE):
                passes = [L.PART_FINE, L.PART_COARSE]

            def note(model, name, samples):
                if model in auto and lib.plan_bwd_compaction(model._plan) in (1, 2, 4):
                    off, nb = C.c_int64(), C.c_int64()
                    lib.render_workspace_region(model_c._plan, plan_f, C.byref(cfg), n, 2, name, C.byref(off), C.byref(nb))
                    so = (off.value + lib.plan_bwd_stats_offset(model._plan, n * samples)) // 4
                    model._auto_note_stats(ws.view(torch.int32)[so:so + 2])

            try:
                with L.launch_on(rays, ws, gpc, gpf, tmp, *[k for k in keep if k is not None]) as st:
                    for part in passes:
                        lib.render_bwd_rays(model_c._plan, plan_f, C.byref(cfg), rays.data_ptr(), n,
                                            packed_c.data_ptr(), packed_f.data_ptr() if nf > 0 else None, C.byref(rr), 0, 0,
                                            C.byref(cot), ws.data_ptr(), wsb, gpc.data_ptr(),
                                            gpf.data_ptr() if gpf is not None else None, part | L.PART_SHARED_BWD,
                                            flats[0].data_ptr() if flats is not None else None,
                                            flats[1].data_ptr() if (flats is not None and flats[1] is not None) else None,
                                            tmp.data_ptr() if tmp is not None else None, tmpb,
                                            g_rays.data_ptr() if g_rays is not None else None, st)
                        if part & L.PART_FINE and len(passes) == 2:
                            note(model_f, b"bwd_scratch_fine", cfg.num_coarse + nf)
                    if parts & L.PART_COARSE:   # (the coarse pass ran last: its words are the ones in the shared buffers)
                        note(model_c, b"bwd_scratch_coarse", cfg.num_coarse)
            finally:
                for plan, was, cur in ((model_c._plan, ctx.bwd_modes[0], now[0]), (plan_f, ctx.bwd_modes[1], now[1])):
                    if plan is not None and was != cur:
                        lib.plan_set_bwd_compaction(plan, cur)
        grads = model_c._split_flat(gpc) + (model_f._split_flat(gpf) if nf > 0 else ())
        return (g_rays,) + (None,) * 6 + grads


def _predict_fused(ray_batch, model_coarse, model_fine, opts):
    rays_grad = bool(ray_batch.requires_grad and torch.is_grad_enabled())
    rays = ray_batch.contiguous().float() if rays_grad else ray_batch.detach().contiguous().float()
    n = rays.shape[0]
    dev = rays.device
    nc, nf = opts.num_coarse, opts.num_fine
    perturb, noise_std = opts.perturb, opts.radiance_field_noise_std
    # the reference's draw order per ray chunk (train_utils.py:63, volume_rendering_utils.py:30, nerf_helpers.py:279,
    # volume_rendering_utils.py:30)
    t_rand = torch.rand((n, nc), dtype=torch.float32, device=dev) if perturb else None
    noise_c = torch.randn((n, nc), dtype=torch.float32, device=dev) if noise_std > 0.0 else None
    u = torch.rand((n, nf), dtype=torch.float32, device=dev) if (nf > 0 and not (perturb == 0.0)) else None
    noise_f = torch.randn((n, nc + nf), dtype=torch.float32, device=dev) if (nf > 0 and noise_std > 0.0) else None
    cfg_tuple = (nc, nf, bool(perturb), bool(opts.lindisp), bool(opts.white_background), float(noise_std))
    pc = model_coarse._ordered_params()
    pf = model_fine._ordered_params() if nf > 0 else []
    training = torch.is_grad_enabled() and (rays_grad or any(p.requires_grad for p in pc + pf))
    outs = _FusedRender.apply(rays, model_coarse, model_fine if nf > 0 else None, cfg_tuple,
                              (t_rand, noise_c, u, noise_f), training, rays_grad, *pc, *pf)
    rgb_c, disp_c, acc_c, depth_c, rgb_f, disp_f, acc_f, depth_f = outs
    if rgb_c.requires_grad:
        # disparity re-derived from the differentiable depth / accumulation maps with the reference's own torch ops
        # (volume_rendering_utils.py:46-48: same values as the kernel's, and autograd covers a loss on it)
        disp_c = 1.0 / torch.max(1e-10 * torch.ones_like(depth_c), depth_c / acc_c)
        if nf > 0:
            disp_f = 1.0 / torch.max(1e-10 * torch.ones_like(depth_f), depth_f / acc_f)
    if nf > 0:
        return rgb_c, disp_c, acc_c, rgb_f, disp_f, acc_f
    return rgb_c, disp_c, acc_c, None, None, None


def predict_and_render_radiance(ray_batch, model_coarse, model_fine, options, mode="train", encode_position_fn=None,
                                encode_direction_fn=None):
    """nerf/train_utils.py:28-127.  Returns (rgb_coarse, disp_coarse, acc_coarse, rgb_fine, disp_fine, acc_fine)."""
    if not ray_batch.is_cuda:
        raise RuntimeError("predict_and_render_radiance needs CUDA (HIP) tensors: nerf_pytorch_amd has no CPU path")
    opts = getattr(options.nerf, mode)
    if _fusable(model_coarse, model_fine, encode_position_fn, encode_direction_fn, opts.num_fine):
        return _predict_fused(ray_batch, model_coarse, model_fine, opts)

    # generic composition (arbitrary networks / encoders), mirroring the reference step by step
    lib = L.get_lib()
    num_rays = ray_batch.shape[0]
    rays = ray_batch.detach().contiguous().float()
    ro, rd = rays[..., :3], rays[..., 3:6]
    dev = rays.device
    nc = opts.num_coarse
    t_rand = torch.rand((num_rays, nc), dtype=torch.float32, device=dev) if opts.perturb else None
    z_vals = torch.empty((num_rays, nc), dtype=torch.float32, device=dev)
    with L.launch_on(rays, t_rand, z_vals) as st:
        lib.stratified_z(rays.data_ptr(), rays.shape[1], num_rays, linspace01(nc, dev).data_ptr(), nc,
                         int(bool(opts.lindisp)), int(bool(opts.perturb)), t_rand.data_ptr() if t_rand is not None else None,
                         0, 0, z_vals.data_ptr(), st)
    pts = ro[..., None, :] + rd[..., None, :] * z_vals[..., :, None]
    radiance_field = run_network(model_coarse, pts, ray_batch, opts.chunksize, encode_position_fn, encode_direction_fn)
    rgb_coarse, disp_coarse, acc_coarse, weights, _ = volume_render_radiance_field(
        radiance_field, z_vals, rd, radiance_field_noise_std=opts.radiance_field_noise_std,
        white_background=opts.white_background)
    rgb_fine, disp_fine, acc_fine = None, None, None
    if opts.num_fine > 0:
        z_vals_mid = 0.5 * (z_vals[..., 1:] + z_vals[..., :-1])
        z_samples = sample_pdf(z_vals_mid, weights[..., 1:-1].detach(), opts.num_fine, det=(opts.perturb == 0.0))
        z_vals, _ = torch.sort(torch.cat((z_vals, z_samples), dim=-1), dim=-1)
        pts = ro[..., None, :] + rd[..., None, :] * z_vals[..., :, None]
        radiance_field = run_network(model_fine, pts, ray_batch, opts.chunksize, encode_position_fn, encode_direction_fn)
        rgb_fine, disp_fine, acc_fine, _, _ = volume_render_radiance_field(
            radiance_field, z_vals, rd, radiance_field_noise_std=opts.radiance_field_noise_std,
            white_background=opts.white_background)
    return rgb_coarse, disp_coarse, acc_coarse, rgb_fine, disp_fine, acc_fine


class _PackRays(torch.autograd.Function):
    """rows [o d near far (v/||v||)] (train_utils.py:143-168) with the gradient autograd gives the reference:
    d(rays)/d(o) = I on columns 0..2, d/d(d) = I on columns 3..5, and the normalisation of the viewdirs columns,
    (g_v - u (u . g_v)) / ||v|| with u = v/||v||, w.r.t. the direction v the reference normalises -- the ray direction itself
    (blender branch) or the PRE-ndc direction (LLFF branch: `vsrc` is a different tensor from `rd` then)."""

    @staticmethod
    def forward(ctx, ro, rd, vsrc, near, far, use_view):
        n = rd.shape[0]
        rays = torch.empty((n, 11 if use_view else 8), dtype=torch.float32, device=rd.device)
        with L.launch_on(ro, rd, vsrc, rays) as st:
            L.get_lib().pack_rays(ro.data_ptr(), rd.data_ptr(), vsrc.data_ptr() if use_view else None, near, far, n,
                                  rays.data_ptr(), st)
        ctx.save_for_backward(vsrc)
        ctx.use_view = use_view
        return rays

    @staticmethod
    def backward(ctx, g):
        (vsrc,) = ctx.saved_tensors
        g_ro, g_rd, g_v = g[:, 0:3].contiguous(), g[:, 3:6].contiguous(), None
        if ctx.use_view:
            nrm = vsrc.norm(p=2, dim=-1, keepdim=True)
            u, gv = vsrc / nrm, g[:, 8:11]
            g_v = (gv - u * (u * gv).sum(-1, keepdim=True)) / nrm
        return g_ro, g_rd, g_v, None, None, None


def pack_rays(ray_origins, ray_directions, options, height=None, width=None, focal_length=None):
    """The ray packing of run_one_iter_of_nerf (nerf/train_utils.py:143-168): rows [o d near far (d/||d||)].
    Differentiable w.r.t. origins and directions (both branches, ndc_rays included) -- what pose optimisation needs."""
    lib = L.get_lib()
    want_grad = torch.is_grad_enabled() and (ray_origins.requires_grad or ray_directions.requires_grad)
    use_view = bool(options.nerf.use_viewdirs)
    if want_grad:
        ro = ray_origins.reshape(-1, 3).contiguous().float()
        rd_src = ray_directions.reshape(-1, 3).contiguous().float()
        rd = rd_src
        if options.dataset.no_ndc is False:  # (train_utils.py:156-160; viewdirs come from the pre-ndc directions: :146-150)
            ro, rd = ndc_rays(height, width, focal_length, 1.0, ro, rd_src)
        return _PackRays.apply(ro, rd, rd_src, float(options.dataset.near), float(options.dataset.far), use_view)
    rd_src = ray_directions.detach().reshape(-1, 3).contiguous().float()
    ro = ray_origins.detach().reshape(-1, 3).contiguous().float()
    rd = rd_src
    if options.dataset.no_ndc is False:
        ro, rd = ndc_rays(height, width, focal_length, 1.0, ro, rd_src)
    n = rd.shape[0]
    rays = torch.empty((n, 11 if use_view else 8), dtype=torch.float32, device=rd.device)
    with L.launch_on(ro, rd, rd_src, rays) as st:
        lib.pack_rays(ro.data_ptr(), rd.data_ptr(), rd_src.data_ptr() if use_view else None, float(options.dataset.near),
                      float(options.dataset.far), n, rays.data_ptr(), st)
    return rays


def _select_cfg(height, width, focal_length, options, channels, seed, step, first):
    f = float(focal_length)
    f32 = lambda v: torch.tensor(v, dtype=torch.float32).item()  # noqa: E731
    return L.SelectCfg(height=int(height), width=int(width), focal=f, near=float(options.dataset.near),
                       far=float(options.dataset.far), use_viewdirs=int(bool(options.nerf.use_viewdirs)),
                       ndc=int(options.dataset.no_ndc is False), ndc_near=1.0, ndc_cw=f32(-1.0 / (width / (2.0 * f))),
                       ndc_ch=f32(-1.0 / (height / (2.0 * f))), ndc_two_near=2.0, ndc_neg_two_near=-2.0,
                       channels=int(channels), seed=int(seed) & 0xFFFFFFFFFFFFFFFF, step=int(step), first=int(first))


def select_training_rays(height, width, focal_length, pose, image, num_random_rays, options, select_inds=None, seed=0,
                         step=0, first=0):
    """The image branch of the training loop (train_nerf.py:210-227) fused with run_one_iter_of_nerf's ray packing
    (train_utils.py:143-168), in ONE launch: draws `num_random_rays` distinct pixels on the device (or takes the
    reference's `select_inds`, the flat indices it draws with np.random.choice), generates only those rays from `pose`
    (>= 3x4, device), and gathers their targets from `image` (H, W, 3|4).  Returns (rays (N, 8|11) -- feed them to
    predict_and_render_radiance or TrainEngine.step --, target (N, C), select_inds (N,))."""
    lib = L.get_lib()
    pose = pose.detach().float()
    if pose.stride(-1) != 1:
        pose = pose.contiguous()
    dev = pose.device
    n = int(num_random_rays)
    channels = 3 if image is None else image.shape[-1]
    cfg = _select_cfg(height, width, focal_length, options, channels, seed, step, first)
    if image is not None:
        image = image.detach().float().contiguous()
    if select_inds is not None:
        select_inds = torch.as_tensor(select_inds, dtype=torch.int64, device=dev).contiguous()
    rays = torch.empty((n, 11 if cfg.use_viewdirs else 8), dtype=torch.float32, device=dev)
    target = torch.empty((n, channels), dtype=torch.float32, device=dev) if image is not None else None
    used = torch.empty((n,), dtype=torch.int64, device=dev)
    with L.launch_on(pose, image, select_inds, rays) as st:
        lib.select_rays(C.byref(cfg), pose.data_ptr(), pose.stride(-2), image.data_ptr() if image is not None else None,
                        select_inds.data_ptr() if select_inds is not None else None, n, rays.data_ptr(),
                        target.data_ptr() if target is not None else None, used.data_ptr(), st)
    return rays, target, used


def select_cached_training_rays(cache_dict, num_random_rays, options, select_inds=None, seed=0, step=0, first=0):
    """The cached branch (train_nerf.py:175-194): rows of cache_dict["ray_bundle"] (2, ., 3) and of
    cache_dict["target"][..., :3], both already on the device."""
    lib = L.get_lib()
    bundle = cache_dict["ray_bundle"]
    ro = bundle[0].reshape((-1, 3)).float().contiguous()
    rd = bundle[1].reshape((-1, 3)).float().contiguous()
    tgt = cache_dict["target"][..., :3].reshape((-1, 3)).float().contiguous()
    dev, n = ro.device, int(num_random_rays)
    cfg = _select_cfg(cache_dict["height"], cache_dict["width"], cache_dict["focal_length"], options, 3, seed, step, first)
    if select_inds is not None:
        select_inds = torch.as_tensor(select_inds, dtype=torch.int64, device=dev).contiguous()
    rays = torch.empty((n, 11 if cfg.use_viewdirs else 8), dtype=torch.float32, device=dev)
    target = torch.empty((n, 3), dtype=torch.float32, device=dev)
    used = torch.empty((n,), dtype=torch.int64, device=dev)
    with L.launch_on(ro, rd, tgt, select_inds, rays) as st:
        lib.select_cached_rays(C.byref(cfg), ro.data_ptr(), rd.data_ptr(), tgt.data_ptr(), ro.shape[0],
                               select_inds.data_ptr() if select_inds is not None else None, n, rays.data_ptr(),
                               target.data_ptr(), used.data_ptr(), st)
    return rays, target, used


def run_one_iter_of_nerf(height, width, focal_length, model_coarse, model_fine, ray_origins, ray_directions, options,
                         mode="train", encode_position_fn=None, encode_direction_fn=None):
    """nerf/train_utils.py:130-202."""
    if not ray_directions.is_cuda:
        raise RuntimeError("run_one_iter_of_nerf needs CUDA (HIP) tensors: nerf_pytorch_amd has no CPU path")
    restore_shapes = [ray_directions.shape, ray_directions.shape[:-1], ray_directions.shape[:-1]]
    if model_fine:
        restore_shapes += restore_shapes
    rays = pack_rays(ray_origins, ray_directions, options, height, width, focal_length)
    batches = get_minibatches(rays, chunksize=getattr(options.nerf, mode).chunksize)
    pred = [predict_and_render_radiance(batch, model_coarse, model_fine, options,
                                        encode_position_fn=encode_position_fn,
                                        encode_direction_fn=encode_direction_fn) for batch in batches]
    synthesized_images = list(zip(*pred))
    synthesized_images = [torch.cat(image, dim=0) if image[0] is not None else None for image in synthesized_images]
    if mode == "validation":
        synthesized_images = [image.view(shape) if image is not None else None
                              for (image, shape) in zip(synthesized_images, restore_shapes)]
        if model_fine:
            return tuple(synthesized_images)
        return tuple(synthesized_images + [None, None, None])
    return tuple(synthesized_images)
