"""Drop-in for ``nerf/volume_rendering_utils.py``: sigma/alpha compositing on the HIP kernels of libnerfhip.so,
differentiable w.r.t. ``radiance_field`` (closed-form backward kernel)."""
import torch

from ._lib import get_lib, launch_on


class _VolumeRender(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, z, rd, noise, noise_std, white):
        lib = get_lib()
        n, s = z.shape
        dev = raw.device
        rgb = torch.empty((n, 3), dtype=torch.float32, device=dev)
        acc = torch.empty((n,), dtype=torch.float32, device=dev)
        depth = torch.empty((n,), dtype=torch.float32, device=dev)
        weights = torch.empty((n, s), dtype=torch.float32, device=dev)
        with launch_on(raw, z, rd, noise, rgb) as st:
            lib.volume_render_fwd(raw.data_ptr(), z.data_ptr(), rd.data_ptr(), 3, n, s, float(noise_std),
                                  noise.data_ptr() if noise is not None else None, 0, 1, 0, int(bool(white)),
                                  rgb.data_ptr(), None, acc.data_ptr(), weights.data_ptr(), depth.data_ptr(), st)
        ctx.save_for_backward(raw, z, rd, noise if noise is not None else torch.empty(0, device=dev))
        ctx.cfg = (float(noise_std), bool(white), noise is not None)
        ctx.set_materialize_grads(False)  # cotangents of unused outputs arrive as None (NULL for the kernel), not as zeros
        return rgb, acc, weights, depth

    @staticmethod
    def backward(ctx, g_rgb, g_acc, g_weights, g_depth):
        lib = get_lib()
        raw, z, rd, noise = ctx.saved_tensors
        noise_std, white, has_noise = ctx.cfg
        n, s = z.shape

        def ptr(g):
            return None if g is None else g.contiguous().float().data_ptr()

        keep = [None if g is None else g.contiguous().float() for g in (g_rgb, g_depth, g_acc, g_weights)]
        g_raw = torch.empty_like(raw)
        with launch_on(raw, z, rd, g_raw, *keep) as st:
            lib.volume_render_bwd(raw.data_ptr(), z.data_ptr(), rd.data_ptr(), 3, n, s, noise_std,
                                  noise.data_ptr() if has_noise else None, 0, 1, 0, int(white),
                                  *[None if k is None else k.data_ptr() for k in keep], g_raw.data_ptr(), st)
        return g_raw, None, None, None, None, None


def volume_render_radiance_field(radiance_field, depth_values, ray_directions, radiance_field_noise_std=0.0,
                                 white_background=False):
    """nerf/volume_rendering_utils.py:6-53.  Returns (rgb_map, disp_map, acc_map, weights, depth_map).  The sigma noise
    is drawn with torch.randn on the tensors' device exactly where the reference draws it (:29-36)."""
    if not radiance_field.is_cuda:
        raise RuntimeError("volume_render_radiance_field needs CUDA (HIP) tensors: nerf_pytorch_amd has no CPU path")
    lead = depth_values.shape[:-1]
    s = depth_values.shape[-1]
    raw = radiance_field.reshape(-1, s, 4).contiguous().float()
    z = depth_values.detach().reshape(-1, s).contiguous().float()
    rd = ray_directions.detach().reshape(-1, 3).contiguous().float()
    noise = None
    if radiance_field_noise_std > 0.0:
        noise = torch.randn(radiance_field[..., 3].shape, dtype=radiance_field.dtype,
                            device=radiance_field.device).reshape(-1, s).contiguous().float()
    rgb, acc, weights, depth = _VolumeRender.apply(raw, z, rd, noise, radiance_field_noise_std, white_background)
    # disparity from depth and acc in torch (two tiny (N,) ops) so that autograd covers it; NaN where acc == 0 (:48)
    disp = 1.0 / torch.max(1e-10 * torch.ones_like(depth), depth / acc)
    return (rgb.reshape(list(lead) + [3]), disp.reshape(lead), acc.reshape(lead), weights.reshape(list(lead) + [s]),
            depth.reshape(lead))
