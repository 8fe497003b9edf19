"""TrainEngine: one NeRF training iteration (train_nerf.py:229-270) as a fixed sequence of C-ABI calls on one HIP
stream, with no host synchronisation inside the step:

    render_fwd (coarse + fine, in-kernel Philox draws) -> mse loss + cotangents -> render_bwd (both nets)
    -> [RCCL all-reduce of the flat gradient, one collective for both nets] -> fused Adam on the flat parameters
    -> re-pack the MFMA weight images.

Data parallelism (BASELINE config 3): one process per GPU, weights replicated, each rank renders its own N/G rays;
the only exchange is the all-reduce (sum) of the 2 x 595,844-float gradient, scaled by 1/G inside the Adam kernel.
Every rank applies the identical update, so no parameter broadcast is needed after step 0.
"""
import ctypes as C
import math

import torch

from . import _lib as L
from .nerf_helpers import linspace01
from .parallel import allreduce_gradients


def _stream():
    return torch.cuda.current_stream().cuda_stream


class TrainEngine:
    def __init__(self, model_coarse, model_fine, num_coarse, num_fine, perturb=True, lindisp=False, white_background=False,
                 noise_std=0.0, lr=5e-3, betas=(0.9, 0.999), eps=1e-8, seed=0, process_group=None, world_size=None,
                 rank=None):
        self.lib = L.get_lib()
        self.mc, self.mf = model_coarse, model_fine if num_fine > 0 else None
        self.dev = model_coarse.flat_params.device
        if self.dev.type != "cuda":
            raise RuntimeError("TrainEngine needs the models on a CUDA (HIP) device")
        self.view = bool(model_coarse.cfg["use_viewdirs"])
        self.stride = 11 if self.view else 8
        self.cfg = L.RenderCfg(num_coarse, num_fine, int(bool(perturb)), int(bool(lindisp)), int(bool(white_background)),
                               float(noise_std), self.stride)
        self.lr, self.betas, self.eps = lr, betas, eps
        self.seed = seed
        self.step_count = 0
        self.pg = process_group
        if world_size is None:
            world_size = torch.distributed.get_world_size(process_group) if torch.distributed.is_initialized() else 1
        if rank is None:
            rank = torch.distributed.get_rank(process_group) if torch.distributed.is_initialized() else 0
        self.world, self.rank = world_size, rank
        # one flat gradient / Adam-state buffer covering both nets: a single collective per step
        self.nc_params = model_coarse.num_flat_params
        self.nf_params = self.mf.num_flat_params if self.mf is not None else 0
        tot = self.nc_params + self.nf_params
        self.grad = torch.zeros(tot, dtype=torch.float32, device=self.dev)
        self.exp_avg = torch.zeros_like(self.grad)
        self.exp_avg_sq = torch.zeros_like(self.grad)
        self.loss = torch.zeros(3, dtype=torch.float32, device=self.dev)
        self._ws = None
        self._ws_n = -1
        self._bufs = None
        self.t_vals = linspace01(num_coarse, self.dev)
        self.u_det = linspace01(num_fine, self.dev) if num_fine > 0 else None
        self.repack()

    def repack(self):
        self.packed_c = self.mc._packed(True)
        self.packed_f = self.mf._packed(True) if self.mf is not None else None

    def _prepare(self, n):
        if self._ws_n == n:
            return
        lib = self.lib
        plan_f = self.mf._plan if self.mf is not None else None
        wsb = lib.render_workspace_bytes(self.mc._plan, plan_f, C.byref(self.cfg), n, 1)
        if wsb < 0:
            raise L.NerfHipError(lib.last_error().decode())
        self._ws = torch.empty(wsb // 4 + 1, dtype=torch.float32, device=self.dev)
        self._wsb = wsb
        self._ws_n = n
        mk = lambda *s: torch.empty(s, dtype=torch.float32, device=self.dev)  # noqa: E731
        self._bufs = dict(rgb_c=mk(n, 3), rgb_f=mk(n, 3), g_c=mk(n, 3), g_f=mk(n, 3), disp_c=mk(n), acc_c=mk(n),
                          disp_f=mk(n), acc_f=mk(n))

    def workspace_bytes(self):
        return 0 if self._ws is None else self._wsb

    def forward_backward(self, rays, target, ray_offset=0):
        """rays: (n, 8|11) packed rows on the device; target: (n, >=3).  Leaves the summed-over-this-rank gradient in
        self.grad and {coarse_mse, fine_mse, sum} in self.loss (device)."""
        lib, n = self.lib, rays.shape[0]
        self._prepare(n)
        b = self._bufs
        nf = self.cfg.num_fine
        plan_f = self.mf._plan if self.mf is not None else None
        out = L.RenderOut(b["rgb_c"].data_ptr(), b["disp_c"].data_ptr(), b["acc_c"].data_ptr(), None,
                          b["rgb_f"].data_ptr() if nf > 0 else None, b["disp_f"].data_ptr() if nf > 0 else None,
                          b["acc_f"].data_ptr() if nf > 0 else None, None)
        seed = self.seed + self.step_count * 0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFFFFFF
        st = _stream()
        pf = self.packed_f.data_ptr() if nf > 0 else None
        lib.render_fwd(self.mc._plan, plan_f, C.byref(self.cfg), rays.data_ptr(), n, self.packed_c.data_ptr(), pf,
                       self.t_vals.data_ptr(), self.u_det.data_ptr() if nf > 0 else None, None, seed, ray_offset,
                       C.byref(out), self._ws.data_ptr(), self._wsb, 1, st)
        lib.mse_loss_fwd_bwd(b["rgb_c"].data_ptr(), b["rgb_f"].data_ptr() if nf > 0 else None, target.data_ptr(),
                             target.shape[1], n, 1.0, b["g_c"].data_ptr(), b["g_f"].data_ptr() if nf > 0 else None,
                             self.loss.data_ptr(), st)
        gc = self.grad[:self.nc_params]
        gf = self.grad[self.nc_params:] if nf > 0 else None
        lib.render_bwd(self.mc._plan, plan_f, C.byref(self.cfg), rays.data_ptr(), n, self.packed_c.data_ptr(), pf, None,
                       seed, ray_offset, b["g_c"].data_ptr(), b["g_f"].data_ptr() if nf > 0 else None,
                       self._ws.data_ptr(), self._wsb, gc.data_ptr(), gf.data_ptr() if gf is not None else None, st)

    def optimizer_step(self, lr=None):
        lib, st = self.lib, _stream()
        self.step_count += 1
        lr = self.lr if lr is None else lr
        scale = 1.0 / self.world
        b1, b2 = self.betas
        n0 = self.nc_params
        lib.adam_step(self.mc.flat_params.data_ptr(), self.grad.data_ptr(), self.exp_avg.data_ptr(),
                      self.exp_avg_sq.data_ptr(), n0, lr, b1, b2, self.eps, self.step_count, scale, st)
        if self.mf is not None:
            lib.adam_step(self.mf.flat_params.data_ptr(), self.grad[n0:].data_ptr(), self.exp_avg[n0:].data_ptr(),
                          self.exp_avg_sq[n0:].data_ptr(), self.nf_params, lr, b1, b2, self.eps, self.step_count, scale, st)
        self.repack()

    def step(self, rays, target, ray_offset=0, lr=None):
        """One full training iteration.  Returns the device tensor {coarse_mse, fine_mse, sum} (no host sync)."""
        self.forward_backward(rays, target, ray_offset)
        if self.world > 1:
            allreduce_gradients(self.grad, self.pg)
        self.optimizer_step(lr)
        return self.loss

    def step_on_image(self, image, pose, height, width, focal_length, options, num_random_rays, lr=None):
        """One whole iteration of the reference's loop body (train_nerf.py:210-270) on a resident training image:
        on-device selection of this rank's `num_random_rays` distinct pixels (ranks take disjoint slices of one
        permutation keyed by (seed, iteration)), their rays and targets, then `step`.  No host work besides launches."""
        from .train_utils import select_training_rays
        n = int(num_random_rays)
        rays, target, _ = select_training_rays(height, width, focal_length, pose, image, n, options, seed=self.seed,
                                               step=self.step_count, first=self.rank * n)
        return self.step(rays, target, ray_offset=self.rank * n, lr=lr)

    @staticmethod
    def lr_at(iteration, lr0=5e-3, lr_decay=250, lr_decay_factor=0.1):
        """train_nerf.py:264-270: lr0 * factor ** (i / (lr_decay * 1000))."""
        return lr0 * (lr_decay_factor ** (iteration / (lr_decay * 1000.0)))

    @staticmethod
    def psnr(loss_sum):
        v = float(loss_sum)
        return -10.0 * math.log10(v if v != 0 else 1e-5)
