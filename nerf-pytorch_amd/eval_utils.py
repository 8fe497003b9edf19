"""The output stage of ``eval_nerf.py``: 8-bit conversion on the device, asynchronous
device-to-host copies into pinned buffers on a side stream, and PNG encoding on worker threads -- so that the render
loop never waits for an image to be written (eval_nerf.py:23-36, 178-190).

    writer = ImageWriter()
    for i, pose in enumerate(render_poses):
        rgb, disp = ...run_one_iter_of_nerf(..., mode="validation")...
        writer.submit(os.path.join(savedir, f"{i:04d}.png"), rgb[..., :3])
        writer.submit(os.path.join(savedir, "disparity", f"{i:04d}.png"), disp, disparity=True)
    writer.close()
"""
import concurrent.futures
import os
import struct
import zlib

import numpy as np
import torch

from . import _lib as L


def _cast_to_image_device(tensor):
    if not tensor.is_cuda:
        raise RuntimeError("cast_to_image needs a CUDA (HIP) tensor: nerf_pytorch_amd has no CPU path")
    t = tensor.detach().float().contiguous()
    h, w, c = t.shape
    out = torch.empty((h, w, 3), dtype=torch.uint8, device=t.device)
    with L.launch_on(t, out) as st:
        L.get_lib().cast_to_image(t.data_ptr(), c, h * w, out.data_ptr(), st)
    return out


def _cast_to_disparity_device(tensor):
    if not tensor.is_cuda:
        raise RuntimeError("cast_to_disparity_image needs a CUDA (HIP) tensor: nerf_pytorch_amd has no CPU path")
    t = tensor.detach().float().contiguous()
    out = torch.empty(t.shape, dtype=torch.uint8, device=t.device)
    scratch = torch.empty(3, dtype=torch.float32, device=t.device)
    with L.launch_on(t, out, scratch) as st:
        L.get_lib().cast_to_disparity_image(t.data_ptr(), t.numel(), scratch.data_ptr(), out.data_ptr(), st)
    return out


def cast_to_image(tensor, dataset_type=None):
    """eval_nerf.py:23-29: (H, W, 3) float in [0, 1] -> (H, W, 3) uint8 numpy array."""
    return _cast_to_image_device(tensor).cpu().numpy()


def cast_to_disparity_image(tensor):
    """eval_nerf.py:32-35: min-max normalised uint8 image (all zeros when the map holds a NaN, as in the reference)."""
    return _cast_to_disparity_device(tensor).cpu().numpy()


def png_bytes(img):
    """A minimal PNG encoder (8-bit grey or RGB, zlib level 3) -- the reference hands the array to imageio.imwrite."""
    img = np.ascontiguousarray(img, np.uint8)
    if img.ndim == 2:
        img = img[..., None]
    h, w, c = img.shape
    if c not in (1, 3):
        raise ValueError("png_bytes: expected (H, W), (H, W, 1) or (H, W, 3) uint8")
    raw = np.empty((h, 1 + w * c), np.uint8)
    raw[:, 0] = 0  # filter type 0 on every scanline
    raw[:, 1:] = img.reshape(h, w * c)

    def chunk(tag, data):
        body = tag + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xFFFFFFFF)

    head = struct.pack(">IIBBBBB", w, h, 8, 2 if c == 3 else 0, 0, 0, 0)
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", head) + chunk(b"IDAT", zlib.compress(raw.tobytes(), 3)) + chunk(b"IEND", b"")


def render_pose_rows(height, width, focal_length, pose, model_coarse, model_fine, options, encode_position_fn=None,
                     encode_direction_fn=None, rank=0, world_size=1, mode="validation"):
    """One pose of the render loop of eval_nerf.py (:158-176), ray-sharded (BASELINE config 5): rank `rank` of
    `world_size` generates and renders only ITS contiguous block of image rows (parallel.shard_bounds) -- rays are
    independent, so no rank ever needs another rank's data and there is no collective; the concatenation of the ranks'
    blocks in rank order is bit-identical to the image one rank renders (parallel.gather_image_rows does that for a
    writer on rank 0).  Returns (outputs, (row_lo, row_hi)) with `outputs` the 6-tuple of run_one_iter_of_nerf shaped
    (rows, width, .)."""
    from .nerf_helpers import get_rays_at_pixels
    from .parallel import shard_bounds
    from .train_utils import run_one_iter_of_nerf
    lo, hi = shard_bounds(int(height), int(rank), int(world_size))
    pix = torch.arange(lo * int(width), hi * int(width), dtype=torch.int64, device=pose.device)
    ro, rd = get_rays_at_pixels(height, width, focal_length, pose[:3, :4] if pose.shape[0] > 3 else pose, pix)
    shape = (hi - lo, int(width), 3)
    out = run_one_iter_of_nerf(height, width, focal_length, model_coarse, model_fine, ro.view(shape), rd.view(shape), options,
                               mode=mode, encode_position_fn=encode_position_fn, encode_direction_fn=encode_direction_fn)
    return out, (lo, hi)


class ImageWriter:
    """Takes rendered maps off the critical path: the 8-bit cast is queued on the render stream, the D2H copy runs on
    a side stream into a pinned buffer, and encoding + file IO happen on worker threads."""

    def __init__(self, workers=4):
        self._pool = concurrent.futures.ThreadPoolExecutor(max_workers=workers)
        self._copy_stream = None
        self._pending = []

    def submit(self, path, tensor, disparity=False):
        dev8 = _cast_to_disparity_device(tensor) if disparity else _cast_to_image_device(tensor)
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=dev8.device)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(dev8.device))
        host = torch.empty(dev8.shape, dtype=torch.uint8, pin_memory=True)
        with torch.cuda.stream(self._copy_stream):
            self._copy_stream.wait_event(ready)
            host.copy_(dev8, non_blocking=True)
            dev8.record_stream(self._copy_stream)
            done = torch.cuda.Event()
            done.record(self._copy_stream)

        def work():
            done.synchronize()
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            with open(path, "wb") as f:
                f.write(png_bytes(host.numpy()))
            return path

        fut = self._pool.submit(work)
        self._pending.append(fut)
        return fut

    def close(self):
        paths = [f.result() for f in self._pending]
        self._pending = []
        self._pool.shutdown(wait=True)
        return paths
