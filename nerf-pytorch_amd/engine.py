"""TrainEngine: one NeRF training iteration (train_nerf.py:229-270) as a fixed graph of C-ABI calls on two HIP
streams, with no host synchronisation inside the step.  The reference runs everything sequentially
(nerf/train_utils.py:68-117 coarse then fine, train_nerf.py:244-261 loss, backward, optimiser); here

    main stream:  coarse forward --E1--> hierarchical sampling + fine forward -> fine loss -> fine backward
                  -> [all-reduce of the fine net's gradient, async] ------------------------------+
    side stream:           E1 -> coarse loss -> coarse backward --E2-->                           |
    main stream:                                            wait E2 -> [all-reduce coarse] -> wait both -> Adam -> re-pack

the coarse net's backward needs nothing from the fine pass, so it runs next to the fine forward/backward (fills the
tails of those launches), and with G > 1 ranks the fine net's gradient all-reduce (RCCL) is in flight while the coarse
backward still computes.  `overlap=False` gives the single-stream order (coarse+fine forward, loss, fine backward,
coarse backward) with the same collectives -- in both orders the fine net's all-reduce is launched before the coarse
backward.  Default: two streams for nets narrower than 256 (measured +3.5 %), one stream for 256-wide nets (-0.4 %).

Data parallelism (BASELINE config 3): one process per GPU, weights replicated, each rank renders its own N/G rays;
the only exchange is the all-reduce (sum) of the 2 x 595,844-float gradient (one collective per net), scaled by 1/G
inside the Adam kernel.  Every rank applies the identical update, so no parameter broadcast is needed after step 0.
"""
import ctypes as C
import math

import torch

from . import _lib as L
from .nerf_helpers import linspace01
from .parallel import allreduce_gradients


class TrainEngine:
    def __init__(self, model_coarse, model_fine, num_coarse, num_fine, perturb=True, lindisp=False, white_background=False,
                 noise_std=0.0, lr=5e-3, betas=(0.9, 0.999), eps=1e-8, seed=0, process_group=None, world_size=None,
                 rank=None, overlap=None, always_reduce=False, backward=None):
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
        # the gradient collectives are issued when there is somebody to exchange with -- or always (always_reduce: a
        # one-rank group still goes through RCCL's work handles and stream ordering; the single-GPU test of that path)
        self._reduce = world_size > 1 or (bool(always_reduce) and torch.distributed.is_initialized())
        # one flat gradient / Adam-state buffer covering both nets: a single collective per step
        self.nc_params = model_coarse.num_flat_params
        self.nf_params = self.mf.num_flat_params if self.mf is not None else 0
        tot = self.nc_params + self.nf_params
        self.grad = torch.zeros(tot, dtype=torch.float32, device=self.dev)
        self.exp_avg = torch.zeros_like(self.grad)
        self.exp_avg_sq = torch.zeros_like(self.grad)
        self.loss = torch.zeros(3, dtype=torch.float32, device=self.dev)
        self._loss_c = torch.zeros(3, dtype=torch.float32, device=self.dev)
        self._loss_f = torch.zeros(3, dtype=torch.float32, device=self.dev)
        # two-stream graph: measured on MI355X (profiles/r02_overlap_ab.txt) +3.5 % for 128-wide nets (the short kernels'
        # tails fill), -0.4 % for 256-wide fp32 nets (every launch already fills the chip for milliseconds); the fp16-piece
        # plans at 256 wide (HBM- and latency-bound kernels that leave the matrix pipe idle half the time: room for a second
        # stream): +1.4 % dense, +6 % compacted, +7 % recomputed (profiles/r06_bench_trained_overlap1.json against
        # r06_bench_trained.json of round 6's first passes) -> default by width and arithmetic
        self.overlap = ((model_coarse.cfg["hidden_size"] <= 128 or getattr(model_coarse, "training_precision", "fp32") != "fp32")
                        if overlap is None else bool(overlap))
        self._side = None       # second HIP stream of this device (created on first use)
        self._ev = None
        self._pending = []      # in-flight gradient all-reduces of the current step
        self._ws = None
        self._ws_n = -1
        self._bufs = None
        # backward mode of the step: None -- whatever each model's set_backward_compaction says (default: dense); "dense" / "compact" /
        # "recompute" -- set on both models; "auto" -- chosen per net and per step from the zero-cotangent fraction the previous
        # compacted steps reported (read back asynchronously: no host synchronisation), see _choose_backward_modes
        if backward not in (None, "dense", "compact", "recompute", "fused", "fused_compact", "fused_stash", "auto"):
            raise ValueError("backward must be None, 'dense', 'compact', 'recompute', 'fused', 'fused_compact', 'fused_stash' or 'auto' (got %r)" % (backward,))
        self.backward = backward
        self._zero_frac = {"coarse": None, "fine": None}   # last known fraction of all-zero d(loss)/d(raw) rows per net
        self._stats_host = None
        self._stats_event = None
        self._stats_pending = None
        self._probe_every = 50
        # steps run dense / compacted / recomputed / fused / fused over the list / fused over the stash, per net ("auto")
        self.backward_modes_used = {"coarse": [0, 0, 0, 0, 0, 0], "fine": [0, 0, 0, 0, 0, 0]}
        if backward in ("dense", "compact", "recompute", "fused", "fused_compact", "fused_stash"):
            for m in (self.mc, self.mf):
                if m is not None:
                    m.set_backward_compaction({"dense": False, "compact": True}.get(backward, backward))
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

    def _check_inputs(self, rays, target):
        for name, t in (("rays", rays), ("target", target)):
            if not isinstance(t, torch.Tensor) or not t.is_cuda or t.device != self.dev:
                raise RuntimeError("TrainEngine: %s must be a tensor on %s (nerf_pytorch_amd has no CPU path)" % (name, self.dev))
            if t.dtype != torch.float32:
                raise RuntimeError("TrainEngine: %s must be float32 (got %s)" % (name, t.dtype))
            if t.dim() != 2:
                raise RuntimeError("TrainEngine: %s must be 2-D (got shape %s)" % (name, tuple(t.shape)))
        if rays.shape[1] != self.stride or not rays.is_contiguous():
            raise RuntimeError("TrainEngine: rays must be contiguous rows of %d floats (got shape %s, strides %s)"
                               % (self.stride, tuple(rays.shape), rays.stride()))
        if target.shape[0] != rays.shape[0] or target.shape[1] < 3 or target.stride(1) != 1:
            # e.g. target_s[..., :3] of an RGBA image is fine (row stride 4 is passed on), a transposed view is not
            raise RuntimeError("TrainEngine: target must hold one row of >= 3 unit-stride floats per ray (got shape %s, "
                               "strides %s)" % (tuple(target.shape), target.stride()))

    def _streams(self):
        main = torch.cuda.current_stream(self.dev)
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.dev)
            self._ev = (torch.cuda.Event(), torch.cuda.Event())
        return main, self._side

    def forward_backward(self, rays, target, ray_offset=0, global_rays=None, draws=None):
        """rays: (n, 8|11) packed rows on the device; target: (n, >=3), row stride free (an RGBA image's [..., :3] view
        works).  Leaves the summed-over-this-rank gradient in self.grad and {coarse_mse, fine_mse, sum} in self.loss
        (device); with world > 1 the gradient all-reduces are in flight when this returns (optimizer_step waits).
        global_rays: total rays of the step over all ranks when the shards are NOT equal -- this rank's cotangents are
        then weighted n * world / global_rays, so that the 1/world-scaled sum is the gradient of the global-batch mean.
        draws: None (production: in-kernel Philox draws keyed by (seed, step, global ray index)) or the reference's four
        draws as device tensors (t_rand (n, nc), noise_coarse (n, nc), u (n, nf), noise_fine (n, nc + nf); any may be
        None) -- e.g. made with torch.rand / torch.randn in the reference's order, to run the engine on exactly the random
        numbers another implementation consumed."""
        self._check_inputs(rays, target)
        lib, n = self.lib, rays.shape[0]
        gscale = 1.0 if global_rays is None else float(n) * self.world / float(global_rays)
        if self.backward == "auto":
            self._choose_backward_modes()
        self._prepare(n)
        b = self._bufs
        nf = self.cfg.num_fine
        plan_f = self.mf._plan if self.mf is not None else None
        out = L.RenderOut(b["rgb_c"].data_ptr(), b["disp_c"].data_ptr(), b["acc_c"].data_ptr(), None,
                          b["rgb_f"].data_ptr() if nf > 0 else None, b["disp_f"].data_ptr() if nf > 0 else None,
                          b["acc_f"].data_ptr() if nf > 0 else None, None)
        seed = self.seed + self.step_count * 0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFFFFFF
        pf = self.packed_f.data_ptr() if nf > 0 else None
        gc = self.grad[:self.nc_params]
        gf = self.grad[self.nc_params:] if nf > 0 else None
        tstride = target.stride(0)
        cot_c = L.RenderCotangents(b["g_c"].data_ptr(), None, None, None, None, None)
        cot_f = L.RenderCotangents(None, None, None, b["g_f"].data_ptr() if nf > 0 else None, None, None)
        rr = None
        if draws is not None:
            shapes = ((n, self.cfg.num_coarse), (n, self.cfg.num_coarse), (n, nf), (n, self.cfg.num_coarse + nf))
            for d, shp in zip(draws, shapes):
                if d is not None and (d.device != self.dev or d.dtype != torch.float32 or tuple(d.shape) != shp or not d.is_contiguous()):
                    raise RuntimeError("TrainEngine: a random-draw tensor must be contiguous float32 %s on %s" % (shp, self.dev))
            self._draws = tuple(draws)  # (kept alive until the next step: the backward kernels read them again)
            rr = C.byref(L.RenderRand(*[None if d is None else d.data_ptr() for d in draws]))
        fwd_args = (self.mc._plan, plan_f, C.byref(self.cfg), rays.data_ptr(), n, self.packed_c.data_ptr(), pf,
                    self.t_vals.data_ptr(), self.u_det.data_ptr() if nf > 0 else None, rr, seed, ray_offset,
                    C.byref(out), self._ws.data_ptr(), self._wsb, 1)
        bwd_head = (self.mc._plan, plan_f, C.byref(self.cfg), rays.data_ptr(), n, self.packed_c.data_ptr(), pf, rr, seed,
                    ray_offset)
        bwd_tail = (self._ws.data_ptr(), self._wsb, gc.data_ptr(), gf.data_ptr() if gf is not None else None)
        self._pending = []
        with torch.cuda.device(self.dev):
            main, side = self._streams()
            st = main.cuda_stream
            two = self.overlap and nf > 0
            lib.render_fwd_parts(*fwd_args, L.PART_COARSE, st)

            def coarse_backward(stream_handle):
                lib.mse_loss_fwd_bwd(b["rgb_c"].data_ptr(), None, target.data_ptr(), tstride, n, gscale, b["g_c"].data_ptr(),
                                     None, self._loss_c.data_ptr(), stream_handle)
                lib.render_bwd_parts(*bwd_head, C.byref(cot_c), *bwd_tail, L.PART_COARSE, stream_handle)

            if two:
                e1, e2 = self._ev
                e1.record(main)
                side.wait_event(e1)
                coarse_backward(side.cuda_stream)
                e2.record(side)
            if nf > 0:
                lib.render_fwd_parts(*fwd_args, L.PART_FINE, st)
                lib.mse_loss_fwd_bwd(b["rgb_f"].data_ptr(), None, target.data_ptr(), tstride, n, gscale, b["g_f"].data_ptr(),
                                     None, self._loss_f.data_ptr(), st)
                lib.render_bwd_parts(*bwd_head, C.byref(cot_f), *bwd_tail, L.PART_FINE, st)
                if self._reduce:  # in flight while the coarse backward computes
                    self._pending.append(allreduce_gradients(gf, self.pg, async_op=True, single_rank=True))
            if two:
                main.wait_event(e2)
            else:
                coarse_backward(st)
            if self._reduce:
                self._pending.append(allreduce_gradients(gc, self.pg, async_op=True, single_rank=True))
            if nf > 0:
                torch.stack((self._loss_c[0], self._loss_f[0], self._loss_c[0] + self._loss_f[0]), out=self.loss)
            else:
                self.loss.copy_(self._loss_c)
            if self.backward == "auto":
                self._request_backward_stats(n)

    # ---- backward="auto" ---------------------------------------------------------------------------------------------------------
    # A compacted step costs what its gather costs when nothing is dropped (fp32: k_wgrad + 19 %, fp16 pieces + 1 %) and saves the
    # dropped fraction of the data and weight gradient; the recomputing mode additionally trades the stash stream of the forward for a
    # second forward over the kept samples (pays above ~2/3 dropped rows for the fp16-piece plans, never for fp32): DESIGN.md 3.3-3.4.
    # Nets with a fused backward (fp32, 64 wide: csrc/mlp64r.hip) always run it -- over every sample (from the register-image stash,
    # mode 5, where the plan has it: 0.70 of the recomputing kernel's time) until the list is known to drop enough of them: 5 % against
    # the recomputing mode 3 (the list costs two small launches), 30 % against mode 5 (the list walk recomputes its forward) --, over
    # the list from there on.
    @staticmethod
    def _mode_for(frac, f16, fused=0):
        if fused:
            dense_mode = 5 if fused == 5 else 3
            return 4 if (frac is not None and frac >= (0.30 if dense_mode == 5 else 0.05)) else dense_mode
        if frac is None:
            return 0
        if f16:
            return 2 if frac >= 0.72 else (1 if frac >= 0.05 else 0)
        return 1 if frac >= 0.15 else 0

    def _choose_backward_modes(self):
        """Sets each net's plan option for the step about to run.  The fractions come from the last compacted step whose two statistics
        words per net have arrived on the host (an asynchronous copy behind that step's backward; polled, never waited for); a net that
        runs dense produces none, so every `_probe_every`-th step runs compacted to look again."""
        if self._stats_event is not None and self._stats_event.query():
            h = self._stats_host.tolist()
            for k, name in enumerate(self._stats_pending):
                kept, total = h[2 * k], h[2 * k + 1]
                if total > 0:
                    self._zero_frac[name] = 1.0 - kept / float(total)
            self._stats_event = None
        probe = self.step_count % self._probe_every == 0
        for name, m in (("coarse", self.mc), ("fine", self.mf)):
            if m is None:
                continue
            mode = self._mode_for(self._zero_frac[name], m.training_precision != "fp32", m.fused_backward_available())
            if probe and mode == 0:
                mode = 1
            if probe and mode in (3, 5):
                mode = 4
            if m.backward_compaction != mode:
                m.set_backward_compaction({0: False, 1: True, 2: "recompute", 3: "fused", 4: "fused_compact", 5: "fused_stash"}[mode])
            self.backward_modes_used[name][mode] += 1

    def _request_backward_stats(self, n):
        """Enqueues the copy of {kept, total} of every net that ran compacted in the step just issued (current stream)."""
        if self._stats_event is not None:
            return  # (the previous request is still in flight)
        names = [nm for nm, m in (("coarse", self.mc), ("fine", self.mf)) if m is not None and m.backward_compaction in (1, 2, 4)]
        if not names:
            return
        if self._stats_host is None:
            self._stats_host = torch.zeros(4, dtype=torch.int32).pin_memory()
        lib = self.lib
        plan_f = self.mf._plan if self.mf is not None else None
        words = self._ws.view(torch.int32)
        for k, name in enumerate(names):
            model = self.mc if name == "coarse" else self.mf
            samples = self.cfg.num_coarse if name == "coarse" else self.cfg.num_coarse + self.cfg.num_fine
            off, nb = C.c_int64(), C.c_int64()
            lib.render_workspace_region(self.mc._plan, plan_f, C.byref(self.cfg), n, 1, ("bwd_scratch_" + name).encode(), C.byref(off), C.byref(nb))
            so = (off.value + lib.plan_bwd_stats_offset(model._plan, n * samples)) // 4
            self._stats_host[2 * k:2 * k + 2].copy_(words[so:so + 2], non_blocking=True)
        self._stats_pending = names
        self._stats_event = torch.cuda.Event()
        self._stats_event.record(torch.cuda.current_stream(self.dev))

    def backward_sample_counts(self):
        """{"coarse": (kept, total), "fine": (kept, total)}: the sample points the last step's COMPACTED backward of each net kept
        (those whose d(loss)/d(raw) row is not all zero) and the sample points of that launch; None for a net whose
        set_backward_compaction is off.  Reads two words per net from the step's workspace: synchronises the device."""
        out = {"coarse": None, "fine": None}
        if self._ws is None:
            return out
        lib, n = self.lib, self._ws_n
        plan_f = self.mf._plan if self.mf is not None else None
        torch.cuda.synchronize(self.dev)
        words = self._ws.view(torch.int32)
        for name, model, samples in (("coarse", self.mc, self.cfg.num_coarse), ("fine", self.mf, self.cfg.num_coarse + self.cfg.num_fine)):
            if model is None or model.backward_compaction not in (1, 2, 4):  # (3: the fused backward over every sample builds no list)
                continue
            off, nb = C.c_int64(), C.c_int64()
            lib.render_workspace_region(self.mc._plan, plan_f, C.byref(self.cfg), n, 1, ("bwd_scratch_" + name).encode(), C.byref(off), C.byref(nb))
            so = lib.plan_bwd_stats_offset(model._plan, n * samples)
            w = words[(off.value + so) // 4:(off.value + so) // 4 + 2].cpu()
            out[name] = (int(w[0]), int(w[1]))
        return out

    def wait_gradients(self):
        """Orders this device's current stream behind the step's gradient all-reduces (no-op without collectives).

        Invariant the step relies on (nccl == RCCL): an asynchronous collective runs on the backend's own stream, which
        first waits for everything enqueued on the stream that was current when it was ISSUED (so the all-reduce of a
        net's gradient starts after that net's k_wgrad_reduce), and `work.wait()` does not block the host: it makes the
        stream that is current when it is CALLED wait for the collective.  Both the issue (forward_backward) and the wait
        happen with self.dev current and on self.dev's current stream -- the stream the Adam kernels are then launched on
        (optimizer_step) -- so Adam reads all-reduced gradients; the next step's kernels follow Adam on the same stream.
        gloo (the CPU / one-device tests) blocks the host in wait() instead: stronger, same result."""
        if not self._pending:
            return
        with torch.cuda.device(self.dev):
            for w in self._pending:
                if w is not None:
                    w.wait()
        self._pending = []

    def collective_times_ms(self, reps=20, warmup=3):
        """Diagnostic for the first real N-GPU run: what ONE gradient all-reduce of each net costs by itself -- the same
        buffers, the same process group, nothing else on the device -- as HIP events on the current stream around a
        synchronous collective (mean of `reps` after `warmup`).  In the step the fine net's all-reduce overlaps the coarse
        backward, so these are upper bounds of what the exchange adds.  Returns {"fine": ms, "coarse": ms} (None entries
        when there is no process group).  The step's pending collectives are waited for first and the live gradient is not
        touched (the transfers run on a zero-filled buffer of the same size)."""
        if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
            return dict(fine=None, coarse=None)
        self.wait_gradients()  # (never next to the step's own collectives: they would reduce the same process group concurrently)
        out = {}
        # a zero-filled scratch of the gradient's size, NOT the live gradient: `warmup + reps` in-place SUM all-reduces multiply a
        # buffer by world^23 (inf / NaN on larger worlds); zeros stay zeros, and the transfer does not depend on the values
        scratch = torch.zeros_like(self.grad)
        with torch.cuda.device(self.dev):
            for name, sl in (("fine", scratch[self.nc_params:]), ("coarse", scratch[:self.nc_params])):
                if sl.numel() == 0:
                    out[name] = None
                    continue
                for _ in range(warmup):
                    allreduce_gradients(sl, self.pg, single_rank=True)
                torch.cuda.synchronize(self.dev)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(reps):
                    allreduce_gradients(sl, self.pg, single_rank=True)
                b.record()
                torch.cuda.synchronize(self.dev)
                out[name] = round(a.elapsed_time(b) / reps, 4)
        return out

    def optimizer_step(self, lr=None):
        lib = self.lib
        self.wait_gradients()
        self.step_count += 1
        lr = self.lr if lr is None else lr
        scale = 1.0 / self.world
        b1, b2 = self.betas
        n0 = self.nc_params
        with L.launch_on(self.grad, self.mc.flat_params) as st:
            lib.adam_step(self.mc.flat_params.data_ptr(), self.grad.data_ptr(), self.exp_avg.data_ptr(),
                          self.exp_avg_sq.data_ptr(), n0, lr, b1, b2, self.eps, self.step_count, scale, st)
            if self.mf is not None:
                lib.adam_step(self.mf.flat_params.data_ptr(), self.grad[n0:].data_ptr(), self.exp_avg[n0:].data_ptr(),
                              self.exp_avg_sq[n0:].data_ptr(), self.nf_params, lr, b1, b2, self.eps, self.step_count, scale,
                              st)
        self.repack()

    def step(self, rays, target, ray_offset=0, lr=None, global_rays=None, draws=None):
        """One full training iteration.  Returns the device tensor {coarse_mse, fine_mse, sum} (no host sync)."""
        self.forward_backward(rays, target, ray_offset, global_rays, draws)
        self.optimizer_step(lr)
        return self.loss

    def step_on_image(self, image, pose, height, width, focal_length, options, num_random_rays, lr=None, global_rays=None):
        """One whole iteration of the reference's loop body (train_nerf.py:210-270) on a resident training image:
        on-device selection of this rank's distinct pixels (ranks take disjoint slices of one permutation keyed by
        (seed, iteration)), their rays and targets, then `step`.  No host work besides launches.
        Weak scaling (default): every rank draws `num_random_rays` rays, the step covers world * num_random_rays.
        Strong scaling (`global_rays` = the step's total, BASELINE config 3: 8192 over 8 ranks): this rank takes its
        parallel.shard_bounds slice of the first `global_rays` positions; unequal shards are weighted (forward_backward)."""
        from .parallel import shard_bounds
        from .train_utils import select_training_rays
        if global_rays is None:
            n = int(num_random_rays)
            first = self.rank * n
        else:
            first, hi = shard_bounds(int(global_rays), self.rank, self.world)
            n = hi - first
        rays, target, _ = select_training_rays(height, width, focal_length, pose, image, n, options, seed=self.seed,
                                               step=self.step_count, first=first)
        return self.step(rays, target, ray_offset=first, lr=lr, global_rays=global_rays)

    @staticmethod
    def lr_at(iteration, lr0=5e-3, lr_decay=250, lr_decay_factor=0.1):
        """train_nerf.py:264-270: lr0 * factor ** (i / (lr_decay * 1000))."""
        return lr0 * (lr_decay_factor ** (iteration / (lr_decay * 1000.0)))

    @staticmethod
    def psnr(loss_sum):
        v = float(loss_sum)
        return -10.0 * math.log10(v if v != 0 else 1e-5)
