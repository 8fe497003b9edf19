"""nerf_oracle.py -- TEST INFRASTRUCTURE.  CPU restatement of the krrish94/nerf-pytorch render + training hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module; the
product package (``nerf-pytorch_amd/``) never does.  It is the checker, never the thing measured or shipped.

Every function restates one reference function (file:line given relative to the reference tree) with plain torch
fp32 CPU ops, in the reference's operation order, but with every random draw turned into an explicit argument
(``t_rand``, ``noise``, ``u``) so that the HIP path can be fed identical draws.

Pinning: ``oracle/gen_golden.py`` imports the real reference (in the build container only) and writes
``tests/golden/*.npz``; ``tests/test_oracle.py`` checks this module against those vectors and against the
known-answer vectors of SURVEY.md Appendix B.  Parity status: pinned against reference-generated goldens (the
reference itself ships no tests for this path -- SURVEY.md section 4).
"""
import math

import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------------------------
# rays
# ---------------------------------------------------------------------------------------------------------------
def get_ray_bundle(height, width, focal, c2w):
    """nerf/nerf_helpers.py:67-110 (+ meshgrid_xy :28-40).  Returns (ray_origins, ray_directions), each (H, W, 3)."""
    cols = torch.arange(width, dtype=c2w.dtype, device=c2w.device)
    rows = torch.arange(height, dtype=c2w.dtype, device=c2w.device)
    ii = cols[None, :].expand(height, width)  # x pixel coordinate varies along the last axis
    jj = rows[:, None].expand(height, width)
    d_cam = torch.stack([(ii - width * 0.5) / focal, -(jj - height * 0.5) / focal, -torch.ones_like(ii)], dim=-1)
    rd = torch.sum(d_cam[..., None, :] * c2w[:3, :3], dim=-1)
    ro = c2w[:3, -1].expand(rd.shape)
    return ro, rd


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """nerf/nerf_helpers.py:170-197."""
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    o = rays_o + t[..., None] * rays_d
    cw = -1.0 / (W / (2.0 * focal))
    ch = -1.0 / (H / (2.0 * focal))
    o0 = cw * o[..., 0] / o[..., 2]
    o1 = ch * o[..., 1] / o[..., 2]
    o2 = 1.0 + 2.0 * near / o[..., 2]
    d0 = cw * (rays_d[..., 0] / rays_d[..., 2] - o[..., 0] / o[..., 2])
    d1 = ch * (rays_d[..., 1] / rays_d[..., 2] - o[..., 1] / o[..., 2])
    d2 = -2.0 * near / o[..., 2]
    return torch.stack([o0, o1, o2], -1), torch.stack([d0, d1, d2], -1)


def pack_rays(ro, rd, near, far, viewdir_src=None):
    """nerf/train_utils.py:143-168: rows [o d near far (viewdir)], viewdir = d / ||d|| of the pre-NDC directions."""
    ro = ro.reshape(-1, 3)
    rd = rd.reshape(-1, 3)
    cols = [ro, rd, near * torch.ones_like(rd[..., :1]), far * torch.ones_like(rd[..., :1])]
    if viewdir_src is not None:
        v = viewdir_src.reshape(-1, 3)
        cols.append(v / v.norm(p=2, dim=-1).unsqueeze(-1))
    return torch.cat(cols, dim=-1)


# ---------------------------------------------------------------------------------------------------------------
# positional encoding
# ---------------------------------------------------------------------------------------------------------------
def frequency_bands(num_freqs, log_sampling=True, dtype=torch.float32):
    """nerf/nerf_helpers.py:133-149."""
    if log_sampling:
        return 2.0 ** torch.linspace(0.0, num_freqs - 1, num_freqs, dtype=dtype)
    return torch.linspace(2.0 ** 0.0, 2.0 ** (num_freqs - 1), num_freqs, dtype=dtype)


def positional_encoding(x, num_freqs=6, include_input=True, log_sampling=True):
    """nerf/nerf_helpers.py:113-157: [x | sin(f0 x) | cos(f0 x) | sin(f1 x) | ...] (blocks of the full last dim)."""
    parts = [x] if include_input else []
    for f in frequency_bands(num_freqs, log_sampling, x.dtype).to(x.device):
        parts.append(torch.sin(x * f))
        parts.append(torch.cos(x * f))
    return parts[0] if len(parts) == 1 else torch.cat(parts, dim=-1)


# ---------------------------------------------------------------------------------------------------------------
# depth samples
# ---------------------------------------------------------------------------------------------------------------
def stratified_z(near, far, num_coarse, lindisp=False, perturb=True, t_rand=None):
    """nerf/train_utils.py:38-65.  near/far: (N,1).  t_rand: (N, num_coarse) uniform draws (needed iff perturb)."""
    t = torch.linspace(0.0, 1.0, num_coarse, dtype=near.dtype).to(near.device)
    if not lindisp:
        z = near * (1.0 - t) + far * t
    else:
        z = 1.0 / (1.0 / near * (1.0 - t) + 1.0 / far * t)
    z = z.expand([near.shape[0], num_coarse])
    if perturb:
        mids = 0.5 * (z[..., 1:] + z[..., :-1])
        upper = torch.cat((mids, z[..., -1:]), dim=-1)
        lower = torch.cat((z[..., :1], mids), dim=-1)
        z = lower + (upper - lower) * t_rand
    return z


# ---------------------------------------------------------------------------------------------------------------
# compositing
# ---------------------------------------------------------------------------------------------------------------
def cumprod_exclusive(x):
    """nerf/nerf_helpers.py:43-64."""
    c = torch.cumprod(x, -1)
    c = torch.roll(c, 1, -1)
    c[..., 0] = 1.0
    return c


def volume_render(raw, z, rd, noise_std=0.0, noise=None, white_background=False):
    """nerf/volume_rendering_utils.py:6-53.  `noise`: N(0,1) draws shaped like z (used iff noise_std > 0).
    Returns (rgb_map, disp_map, acc_map, weights, depth_map)."""
    far_gap = torch.tensor([1e10], dtype=rd.dtype, device=rd.device).expand(z[..., :1].shape)
    dists = torch.cat((z[..., 1:] - z[..., :-1], far_gap), dim=-1)
    dists = dists * rd[..., None, :].norm(p=2, dim=-1)
    rgb = torch.sigmoid(raw[..., :3])
    nz = 0.0
    if noise_std > 0.0:
        nz = noise * noise_std
    sigma_a = F.relu(raw[..., 3] + nz)
    alpha = 1.0 - torch.exp(-sigma_a * dists)
    weights = alpha * cumprod_exclusive(1.0 - alpha + 1e-10)
    rgb_map = (weights[..., None] * rgb).sum(dim=-2)
    depth_map = (weights * z).sum(dim=-1)
    acc_map = weights.sum(dim=-1)
    disp_map = 1.0 / torch.max(1e-10 * torch.ones_like(depth_map), depth_map / acc_map)
    if white_background:
        rgb_map = rgb_map + (1.0 - acc_map[..., None])
    return rgb_map, disp_map, acc_map, weights, depth_map


# ---------------------------------------------------------------------------------------------------------------
# importance sampling
# ---------------------------------------------------------------------------------------------------------------
def sample_pdf(bins, weights, num_samples, det=False, u=None, return_aux=False):
    """nerf/nerf_helpers.py:260-302 (sample_pdf_2, the one the hot path imports -- train_utils.py:4) with the
    torchsearchsorted call (:288) restated as torch.searchsorted(right=True).  `u`: (N, num_samples) uniform draws
    (needed iff not det)."""
    w = weights + 1e-5
    pdf = w / torch.sum(w, dim=-1, keepdim=True)
    cdf = torch.cumsum(pdf, dim=-1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], dim=-1)
    if det:
        u = torch.linspace(0.0, 1.0, steps=num_samples, dtype=w.dtype).to(w.device).expand(list(cdf.shape[:-1]) + [num_samples])
    u = u.contiguous()
    cdf = cdf.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    c0 = torch.gather(cdf, -1, below)
    c1 = torch.gather(cdf, -1, above)
    b0 = torch.gather(bins, -1, below)
    b1 = torch.gather(bins, -1, above)
    denom = c1 - c0
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - c0) / denom
    samples = b0 + t * (b1 - b0)
    if return_aux:
        return samples, inds, cdf
    return samples


def hierarchical_z(z_coarse, weights, num_fine, det=False, u=None):
    """nerf/train_utils.py:96-105: mids, sample_pdf on weights[...,1:-1], detach, sort(cat)."""
    z_mid = 0.5 * (z_coarse[..., 1:] + z_coarse[..., :-1])
    z_samples = sample_pdf(z_mid, weights[..., 1:-1], num_fine, det=det, u=u).detach()
    z_fine, _ = torch.sort(torch.cat((z_coarse, z_samples), dim=-1), dim=-1)
    return z_samples, z_fine


# ---------------------------------------------------------------------------------------------------------------
# the MLP
# ---------------------------------------------------------------------------------------------------------------
def model_dims(cfg):
    dx = (3 if cfg.get("include_input_xyz", True) else 0) + 6 * cfg["num_encoding_fn_xyz"]
    dd = (3 if cfg.get("include_input_dir", True) else 0) + 6 * cfg["num_encoding_fn_dir"]
    if not cfg.get("use_viewdirs", True):
        dd = 0
    return dx, dd


def is_skip_layer(i, cfg):
    """cat(h, xyz) before layers_xyz[i] iff i % skip == 0 and i > 0 (nerf/models.py:208-214, 240-245)."""
    return i % cfg["skip_connect_every"] == 0 and i > 0


def param_shapes(cfg):
    """Tensors of FlexibleNeRFModel in registration order (nerf/models.py:205-229) -> list of (name, shape)."""
    W, L = cfg["hidden_size"], cfg["num_layers"]
    dx, dd = model_dims(cfg)
    out = [("layer1.weight", (W, dx)), ("layer1.bias", (W,))]
    for i in range(L - 1):
        k = W + (dx if is_skip_layer(i, cfg) else 0)
        out += [("layers_xyz.%d.weight" % i, (W, k)), ("layers_xyz.%d.bias" % i, (W,))]
    if cfg.get("use_viewdirs", True):
        out += [("layers_dir.0.weight", (W // 2, W + dd)), ("layers_dir.0.bias", (W // 2,)),
                ("fc_alpha.weight", (1, W)), ("fc_alpha.bias", (1,)),
                ("fc_rgb.weight", (3, W // 2)), ("fc_rgb.bias", (3,)),
                ("fc_feat.weight", (W, W)), ("fc_feat.bias", (W,))]
    else:
        out += [("fc_out.weight", (4, W)), ("fc_out.bias", (4,))]
    return out


def init_params(cfg, seed=0):
    """torch.nn.Linear default init (kaiming_uniform(a=sqrt 5) weights, U(+-1/sqrt(fan_in)) bias) per tensor."""
    g = torch.Generator().manual_seed(seed)
    params = {}
    shapes = dict(param_shapes(cfg))
    for name, shape in param_shapes(cfg):
        if name.endswith(".weight"):
            bound = 1.0 / math.sqrt(shape[1])
            params[name] = (torch.rand(shape, generator=g) * 2 - 1) * bound
        else:
            fan_in = shapes[name[:-5] + ".weight"][1]
            bound = 1.0 / math.sqrt(fan_in)
            params[name] = (torch.rand(shape, generator=g) * 2 - 1) * bound
    return params


def mlp_forward(params, x, cfg):
    """FlexibleNeRFModel.forward (nerf/models.py:233-256) incl. its quirks: no activation after layer1 (:238), skip
    as cat(h, xyz) (:245), sigma taken from h not feat (:249), raw cat(rgb, sigma) output (:254)."""
    dx, dd = model_dims(cfg)
    L = cfg["num_layers"]
    xyz = x[..., :dx]
    h = F.linear(xyz, params["layer1.weight"], params["layer1.bias"])
    for i in range(L - 1):
        if is_skip_layer(i, cfg):
            h = torch.cat((h, xyz), dim=-1)
        h = F.relu(F.linear(h, params["layers_xyz.%d.weight" % i], params["layers_xyz.%d.bias" % i]))
    if cfg.get("use_viewdirs", True):
        view = x[..., dx:]
        feat = F.relu(F.linear(h, params["fc_feat.weight"], params["fc_feat.bias"]))
        alpha = F.linear(h, params["fc_alpha.weight"], params["fc_alpha.bias"])
        y = F.relu(F.linear(torch.cat((feat, view), dim=-1), params["layers_dir.0.weight"], params["layers_dir.0.bias"]))
        rgb = F.linear(y, params["fc_rgb.weight"], params["fc_rgb.bias"])
        return torch.cat((rgb, alpha), dim=-1)
    return F.linear(h, params["fc_out.weight"], params["fc_out.bias"])


def mlp_relu_margin(params, x, cfg, per_layer=False):
    """Per input row: the smallest |pre-activation| among everything FlexibleNeRFModel.forward (nerf/models.py:233-256)
    passes through a ReLU, relative to the largest one of the row.  A row whose margin is ~1e-7 has a unit whose ReLU
    branch is decided by fp32 round-off: two fp32 implementations with different summation orders may take different
    branches there, and that row's gradient then differs by O(1) of the unit's contribution (tests exclude such rows)."""
    dx, dd = model_dims(cfg)
    L = cfg["num_layers"]
    xyz = x[..., :dx]
    h = F.linear(xyz, params["layer1.weight"], params["layer1.bias"])
    pres = []
    for i in range(L - 1):
        if is_skip_layer(i, cfg):
            h = torch.cat((h, xyz), dim=-1)
        pre = F.linear(h, params["layers_xyz.%d.weight" % i], params["layers_xyz.%d.bias" % i])
        pres.append(pre)
        h = F.relu(pre)
    if cfg.get("use_viewdirs", True):
        pre = F.linear(h, params["fc_feat.weight"], params["fc_feat.bias"])
        pres.append(pre)
        pres.append(F.linear(torch.cat((F.relu(pre), x[..., dx:]), dim=-1), params["layers_dir.0.weight"], params["layers_dir.0.bias"]))
    if not pres:
        return torch.ones(x.shape[0])
    if per_layer:  # each layer's smallest |pre-activation| relative to THAT layer's largest (nets whose layers differ by orders of magnitude)
        def one(p):
            mx = p.abs().max(dim=-1).values
            # (a layer whose pre-activations are ALL zero -- a zero input row under zero biases -- has no branch round-off could flip:
            # margin 1, the row is compared; 0 / 0 would be NaN, `NaN > margin` False, and the row silently dropped)
            return torch.where(mx > 0, p.abs().min(dim=-1).values / torch.where(mx > 0, mx, torch.ones_like(mx)), torch.ones_like(mx))
        return torch.stack([one(p) for p in pres], dim=0).min(dim=0).values
    allp = torch.cat(pres, dim=-1)
    return allp.abs().min(dim=-1).values / (allp.abs().max(dim=-1).values + 1e-30)


def run_network(params, pts, rays, cfg, chunksize=None):
    """nerf/train_utils.py:8-25: encode points (+ per-ray view directions broadcast over samples), run the MLP."""
    flat = pts.reshape(-1, 3)
    emb = positional_encoding(flat, cfg["num_encoding_fn_xyz"], cfg.get("include_input_xyz", True),
                              cfg.get("log_sampling_xyz", True))
    if cfg.get("use_viewdirs", True):
        dirs = rays[..., None, -3:].expand(pts.shape).reshape(-1, 3)
        emb_d = positional_encoding(dirs, cfg["num_encoding_fn_dir"], cfg.get("include_input_dir", True),
                                    cfg.get("log_sampling_dir", True))
        emb = torch.cat((emb, emb_d), dim=-1)
    if chunksize is None:
        out = mlp_forward(params, emb, cfg)
    else:
        out = torch.cat([mlp_forward(params, emb[i:i + chunksize], cfg) for i in range(0, emb.shape[0], chunksize)], 0)
    return out.reshape(list(pts.shape[:-1]) + [4])


# ---------------------------------------------------------------------------------------------------------------
# the whole path
# ---------------------------------------------------------------------------------------------------------------
def render_rays(rays, params_c, params_f, cfg_c, cfg_f, opt, rand=None, chunksize=None):
    """predict_and_render_radiance (nerf/train_utils.py:28-127).  `opt`: dict(num_coarse, num_fine, perturb, lindisp,
    white_background, noise_std).  `rand`: dict(t_rand, noise_coarse, u, noise_fine).  Returns a dict with the six
    reference outputs plus depth_*/weights/z_* intermediates."""
    rand = rand or {}
    ro, rd = rays[..., :3], rays[..., 3:6]
    near, far = rays[..., 6:7], rays[..., 7:8]
    z = stratified_z(near, far, opt["num_coarse"], opt.get("lindisp", False), opt.get("perturb", True),
                     rand.get("t_rand"))
    pts = ro[..., None, :] + rd[..., None, :] * z[..., :, None]
    raw_c = run_network(params_c, pts, rays, cfg_c, chunksize)
    rgb_c, disp_c, acc_c, w_c, depth_c = volume_render(raw_c, z, rd, opt.get("noise_std", 0.0), rand.get("noise_coarse"),
                                                       opt.get("white_background", False))
    out = dict(rgb_coarse=rgb_c, disp_coarse=disp_c, acc_coarse=acc_c, depth_coarse=depth_c, weights_coarse=w_c,
               z_coarse=z, raw_coarse=raw_c, rgb_fine=None, disp_fine=None, acc_fine=None, depth_fine=None)
    if opt["num_fine"] > 0:
        det = (opt.get("perturb", True) == 0.0)
        z_samples, z_f = hierarchical_z(z, w_c, opt["num_fine"], det=det, u=rand.get("u"))
        pts_f = ro[..., None, :] + rd[..., None, :] * z_f[..., :, None]
        raw_f = run_network(params_f, pts_f, rays, cfg_f, chunksize)
        rgb_f, disp_f, acc_f, _, depth_f = volume_render(raw_f, z_f, rd, opt.get("noise_std", 0.0),
                                                         rand.get("noise_fine"), opt.get("white_background", False))
        out.update(rgb_fine=rgb_f, disp_fine=disp_f, acc_fine=acc_f, depth_fine=depth_f, z_samples=z_samples, z_fine=z_f,
                   raw_fine=raw_f)
    return out


def loss_and_psnr(rgb_coarse, rgb_fine, target):
    """train_nerf.py:244-260 + mse2psnr (nerf/nerf_helpers.py:13-17): PSNR of the SUMMED coarse+fine mse."""
    lc = F.mse_loss(rgb_coarse[..., :3], target[..., :3])
    lf = F.mse_loss(rgb_fine[..., :3], target[..., :3]) if rgb_fine is not None else None
    loss = lc + lf if lf is not None else lc
    v = float(loss.detach())
    psnr = -10.0 * math.log10(v if v != 0 else 1e-5)
    return loss, lc, lf, psnr


def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam's single-tensor update (what train_nerf.py:141-143,261 runs), in place."""
    m.lerp_(g, 1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


# ---- next to the path: training-ray selection (train_nerf.py:210-227, :175-194) and the 8-bit output stage ----------
def select_training_rays(height, width, focal, c2w, image, select_inds):
    """train_nerf.py:213-227: rays of the whole image, then rows picked through coords[select_inds]."""
    ro, rd = get_ray_bundle(height, width, focal, c2w)
    ii, jj = torch.meshgrid(torch.arange(height), torch.arange(width), indexing="ij")
    coords = torch.stack([ii.transpose(-1, -2), jj.transpose(-1, -2)], dim=-1).reshape((-1, 2))  # meshgrid_xy
    sel = coords[torch.as_tensor(select_inds, dtype=torch.int64)]
    tgt = None if image is None else image[sel[:, 0], sel[:, 1], :]
    return ro[sel[:, 0], sel[:, 1], :], rd[sel[:, 0], sel[:, 1], :], tgt


def select_cached_rays(ray_bundle, target, select_inds):
    """train_nerf.py:175-194: rows of a cached bundle (2, ., 3) and of its targets (first three channels)."""
    idx = torch.as_tensor(select_inds, dtype=torch.int64)
    ro, rd = ray_bundle[0].reshape((-1, 3)), ray_bundle[1].reshape((-1, 3))
    return ro[idx], rd[idx], target[..., :3].reshape((-1, 3))[idx]


def cast_to_image(rgb):
    """eval_nerf.py:23-29.  torchvision (absent here; ToPILImage of a float tensor is pic.mul(255).byte(), functional
    to_pil_image) followed by np.array -> (H, W, 3) uint8."""
    return rgb[..., :3].permute(2, 0, 1).mul(255).byte().permute(1, 2, 0).contiguous().numpy()


def cast_to_disparity_image(t):
    """eval_nerf.py:32-35."""
    import numpy as np
    img = (t - t.min()) / (t.max() - t.min())
    img = img.clamp(0, 1) * 255
    a = img.detach().cpu().numpy()
    # astype(uint8) of NaN is platform-defined; the reference run on x86-64 (tests/golden/dataio.npz) yields 0
    return np.where(np.isnan(a), 0, a).astype(np.uint8)
