"""Drop-in for the reference's ``nerf/nerf_helpers.py``: same names, argument meaning and return layout, but every
function runs a hand-written HIP kernel of libnerfhip.so on the tensors' device (MI355X).  CPU tensors are rejected:
this package has no CPU path.

Functions here are *not* differentiable w.r.t. their tensor inputs (the hot path never needs that: the encodings,
depths and samples carry no gradient -- SURVEY A.8); the differentiable pieces are
``volume_rendering_utils.volume_render_radiance_field`` and ``models.FlexibleNeRFModel``.
"""
import math
from typing import Optional

import torch

from ._lib import get_lib, launch_on


def _dev32(t, what):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("%s must be a CUDA (HIP) tensor: nerf_pytorch_amd has no CPU path" % what)
    if t.dtype != torch.float32:
        raise RuntimeError("%s must be float32 (got %s)" % (what, t.dtype))
    return t.detach().contiguous()


_CONST_CACHE = {}


def linspace01(n, device):
    """torch.linspace(0, 1, n) evaluated on the CPU (the oracle's bits) and cached on `device`."""
    key = ("lin", n, str(device))
    if key not in _CONST_CACHE:
        _CONST_CACHE[key] = torch.linspace(0.0, 1.0, n, dtype=torch.float32).to(device)
    return _CONST_CACHE[key]


def frequency_bands_cpu(num_encoding_functions, log_sampling=True):
    """Frequency bands exactly as nerf/nerf_helpers.py:133-149 builds them (CPU bits)."""
    if log_sampling:
        return 2.0 ** torch.linspace(0.0, num_encoding_functions - 1, num_encoding_functions, dtype=torch.float32)
    return torch.linspace(2.0 ** 0.0, 2.0 ** (num_encoding_functions - 1), num_encoding_functions, dtype=torch.float32)


def _freqs(n, log_sampling, device):
    key = ("freq", n, bool(log_sampling), str(device))
    if key not in _CONST_CACHE:
        _CONST_CACHE[key] = frequency_bands_cpu(n, log_sampling).to(device)
    return _CONST_CACHE[key]


# ---------------------------------------------------------------------------------------------------------------------
def img2mse(img_src, img_tgt):
    """nerf/nerf_helpers.py:9-10."""
    return torch.nn.functional.mse_loss(img_src, img_tgt)


def mse2psnr(mse):
    """nerf/nerf_helpers.py:13-17."""
    if mse == 0:
        mse = 1e-5
    return -10.0 * math.log10(mse)


def get_minibatches(inputs: torch.Tensor, chunksize: Optional[int] = 1024 * 8):
    """nerf/nerf_helpers.py:20-25."""
    return [inputs[i:i + chunksize] for i in range(0, inputs.shape[0], chunksize)]


def meshgrid_xy(tensor1: torch.Tensor, tensor2: torch.Tensor):
    """nerf/nerf_helpers.py:28-40 (numpy "xy" meshgrid)."""
    ii, jj = torch.meshgrid(tensor1, tensor2, indexing="ij")
    return ii.transpose(-1, -2), jj.transpose(-1, -2)


class _CumprodExclusive(torch.autograd.Function):
    """Exclusive cumulative product along the last dimension with the gradient autograd derives for the reference's
    cumprod / roll / overwrite composition (nerf/nerf_helpers.py:43-64) -- tiny_nerf.py:100-101 backpropagates through it."""

    @staticmethod
    def forward(ctx, x):
        cols = x.shape[-1]
        rows = x.numel() // max(cols, 1)
        out = torch.empty_like(x)
        with launch_on(x, out) as st:
            get_lib().cumprod_exclusive(x.data_ptr(), rows, cols, out.data_ptr(), st)
        ctx.save_for_backward(x, out)
        return out

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        cols = x.shape[-1]
        rows = x.numel() // max(cols, 1)
        g = g.contiguous().float()
        gx = torch.empty_like(x)
        with launch_on(x, y, g, gx) as st:
            get_lib().cumprod_exclusive_bwd(x.data_ptr(), y.data_ptr(), g.data_ptr(), rows, cols, gx.data_ptr(), st)
        return gx


def cumprod_exclusive(tensor: torch.Tensor) -> torch.Tensor:
    """nerf/nerf_helpers.py:43-64 -- exclusive cumulative product along the last dimension (differentiable)."""
    if not isinstance(tensor, torch.Tensor) or not tensor.is_cuda:
        raise RuntimeError("tensor must be a CUDA (HIP) tensor: nerf_pytorch_amd has no CPU path")
    if not tensor.is_floating_point():
        raise RuntimeError("tensor must be a floating-point tensor (got %s)" % tensor.dtype)
    # the kernel computes in fp32; other floating dtypes (the reference works for any) are cast in and out, and autograd
    # carries the gradient back through the casts
    return _CumprodExclusive.apply(tensor.float().contiguous()).to(tensor.dtype)


def get_ray_bundle(height: int, width: int, focal_length, tform_cam2world: torch.Tensor):
    """nerf/nerf_helpers.py:67-110.  Returns (ray_origins, ray_directions), each (height, width, 3); directions are
    not normalised."""
    c2w = _dev32(tform_cam2world, "tform_cam2world")
    if c2w.dim() != 2 or c2w.shape[0] < 3 or c2w.shape[1] < 4:
        raise RuntimeError("tform_cam2world must be at least 3x4")
    focal = float(focal_length)
    n = height * width
    ro = torch.empty((height, width, 3), dtype=torch.float32, device=c2w.device)
    rd = torch.empty_like(ro)
    with launch_on(c2w, ro, rd) as st:
        get_lib().ray_bundle(height, width, focal, c2w.data_ptr(), c2w.stride(0), None, n, ro.data_ptr(), rd.data_ptr(), st)
    return ro, rd


def get_rays_at_pixels(height: int, width: int, focal_length, tform_cam2world: torch.Tensor, pixels: torch.Tensor):
    """Rays of selected pixels only: `pixels` are int64 linear ids row*width+col.  Equivalent to
    get_ray_bundle(...)[...].reshape(-1, 3)[pixels] without generating the whole image."""
    c2w = _dev32(tform_cam2world, "tform_cam2world")
    pix = pixels.to(device=c2w.device, dtype=torch.int64).contiguous()
    n = pix.numel()
    ro = torch.empty((n, 3), dtype=torch.float32, device=c2w.device)
    rd = torch.empty_like(ro)
    with launch_on(c2w, pix, ro, rd) as st:
        get_lib().ray_bundle(height, width, float(focal_length), c2w.data_ptr(), c2w.stride(0), pix.data_ptr(), n,
                             ro.data_ptr(), rd.data_ptr(), st)
    return ro, rd


def positional_encoding(tensor, num_encoding_functions=6, include_input=True, log_sampling=True) -> torch.Tensor:
    """nerf/nerf_helpers.py:113-157."""
    x = _dev32(tensor, "tensor")
    if num_encoding_functions == 0 and include_input:
        return tensor
    d = x.shape[-1]
    m = x.numel() // d
    out = torch.empty(list(x.shape[:-1]) + [d * (int(bool(include_input)) + 2 * num_encoding_functions)],
                      dtype=torch.float32, device=x.device)
    fr = _freqs(num_encoding_functions, log_sampling, x.device)
    with launch_on(x, fr, out) as st:
        get_lib().positional_encoding(x.data_ptr(), m, d, fr.data_ptr(), num_encoding_functions, int(bool(include_input)),
                                      out.data_ptr(), st)
    return out


class EmbeddingFunction:
    """What get_embedding_function returns: callable like the reference's lambda, but introspectable so that
    run_network / the fused path can compute the encoding inside the MLP kernel instead of materialising it."""

    def __init__(self, num_encoding_functions, include_input, log_sampling):
        self.num_encoding_functions = num_encoding_functions
        self.include_input = include_input
        self.log_sampling = log_sampling

    def __call__(self, x):
        return positional_encoding(x, self.num_encoding_functions, self.include_input, self.log_sampling)


def get_embedding_function(num_encoding_functions=6, include_input=True, log_sampling=True):
    """nerf/nerf_helpers.py:160-167."""
    return EmbeddingFunction(num_encoding_functions, include_input, log_sampling)


class _NdcRays(torch.autograd.Function):
    """ndc_rays with its vector-Jacobian product (nerfhip_ndc_rays_bwd)."""

    @staticmethod
    def forward(ctx, o, d, consts):
        oo, od = torch.empty_like(o), torch.empty_like(d)
        with launch_on(o, d, oo, od) as st:
            get_lib().ndc_rays(*consts, o.data_ptr(), d.data_ptr(), o.numel() // 3, oo.data_ptr(), od.data_ptr(), st)
        ctx.save_for_backward(o, d)
        ctx.consts = consts
        ctx.set_materialize_grads(False)
        return oo, od

    @staticmethod
    def backward(ctx, g_oo, g_od):
        o, d = ctx.saved_tensors
        if g_oo is None and g_od is None:
            return None, None, None
        g_oo = torch.zeros_like(o) if g_oo is None else g_oo.contiguous().float()
        g_od = torch.zeros_like(d) if g_od is None else g_od.contiguous().float()
        g_o, g_d = torch.empty_like(o), torch.empty_like(d)
        with launch_on(o, d, g_oo, g_od, g_o, g_d) as st:
            get_lib().ndc_rays_bwd(*ctx.consts, o.data_ptr(), d.data_ptr(), g_oo.data_ptr(), g_od.data_ptr(), o.numel() // 3,
                                   g_o.data_ptr(), g_d.data_ptr(), st)
        return g_o, g_d, None


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """nerf/nerf_helpers.py:170-197.  Differentiable w.r.t. rays_o / rays_d, as the reference's tensor arithmetic is."""
    want_grad = torch.is_grad_enabled() and (rays_o.requires_grad or rays_d.requires_grad)
    o = rays_o.contiguous().float() if want_grad else _dev32(rays_o, "rays_o")
    d = rays_d.contiguous().float() if want_grad else _dev32(rays_d, "rays_d")
    if isinstance(focal, torch.Tensor):  # the reference then evaluates the constants in fp32 tensor arithmetic
        f = focal.detach().float().cpu()
        cw = float(-1.0 / (W / (2.0 * f)))
        ch = float(-1.0 / (H / (2.0 * f)))
    else:
        cw = -1.0 / (W / (2.0 * focal))
        ch = -1.0 / (H / (2.0 * focal))
    if want_grad:
        if o.device.type != "cuda":
            raise RuntimeError("ndc_rays: rays_o must live on the GPU (got %s)" % o.device)
        return _NdcRays.apply(o, d, (float(near), cw, ch, 2.0 * near, -2.0 * near))
    n = o.numel() // 3
    oo, od = torch.empty_like(o), torch.empty_like(d)
    with launch_on(o, d, oo, od) as st:
        get_lib().ndc_rays(float(near), cw, ch, 2.0 * near, -2.0 * near, o.data_ptr(), d.data_ptr(), n, oo.data_ptr(),
                           od.data_ptr(), st)
    return oo, od


def sample_pdf_2(bins, weights, num_samples, det=False):
    """nerf/nerf_helpers.py:260-302 (the sampler the hot path uses, train_utils.py:4).  Random draws come from
    torch.rand on the tensors' device, in the reference's order and shape."""
    b = _dev32(bins, "bins")
    w = _dev32(weights, "weights")
    nb = b.shape[-1]
    n = b.numel() // nb
    if w.shape[-1] != nb - 1:
        raise RuntimeError("weights must have one entry fewer than bins")
    u = None
    if not det:
        u = torch.rand(list(w.shape[:-1]) + [num_samples], dtype=torch.float32, device=w.device).contiguous()
    out = torch.empty(list(b.shape[:-1]) + [num_samples], dtype=torch.float32, device=b.device)
    with launch_on(b, w, u, out) as st:
        get_lib().sample_pdf(b.data_ptr(), w.data_ptr(), n, nb, u.data_ptr() if u is not None else None, int(bool(det)),
                             linspace01(num_samples, b.device).data_ptr(), num_samples, 0, 0, out.data_ptr(), None, None, st)
    return out


def sample_pdf(bins, weights, num_samples, det=False):
    """nerf/nerf_helpers.py:222-257 -- the older variant (dead code on the hot path); same results."""
    return sample_pdf_2(bins, weights, num_samples, det)


def sample_pdf_with_indices(bins, weights, u):
    """sample_pdf_2 with caller-supplied uniform draws `u`, also returning the searchsorted(side="right") indices
    (int64, nerf/nerf_helpers.py:288) and the CDF -- the entry point of the bit-exact index parity test."""
    b = _dev32(bins, "bins")
    w = _dev32(weights, "weights")
    uu = _dev32(u, "u")
    nb = b.shape[-1]
    n = b.numel() // nb
    nf = uu.shape[-1]
    out = torch.empty_like(uu)
    inds = torch.empty(uu.shape, dtype=torch.int64, device=uu.device)
    cdf = torch.empty_like(b)
    with launch_on(b, w, uu, out, inds, cdf) as st:
        get_lib().sample_pdf(b.data_ptr(), w.data_ptr(), n, nb, uu.data_ptr(), 0, None, nf, 0, 0, out.data_ptr(),
                             inds.data_ptr(), cdf.data_ptr(), st)
    return out, inds, cdf
