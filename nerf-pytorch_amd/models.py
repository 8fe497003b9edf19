"""Drop-in for ``nerf/models.py``'s FlexibleNeRFModel (nerf/models.py:185-256): same constructor arguments, same
parameter names / shapes / state_dict keys (so reference checkpoints load and save unchanged), same forward contract
``x[M, dim_xyz + dim_dir] -> [M, 4] = cat(rgb_raw, sigma_raw)`` -- but forward and backward are the fp32-MFMA kernels
of libnerfhip.so, and all parameters are views into ONE flat fp32 buffer (what the RCCL gradient all-reduce and the
fused Adam step operate on).

Unlike the reference as shipped, geometries with an active skip connection work (the reference raises AttributeError
at models.py:243 -- SURVEY 0.3); the rule implemented is the one its __init__ encodes: cat(h, xyz) before
layers_xyz[i] iff i % skip_connect_every == 0 and i > 0.
"""
import ctypes as C

import torch

from . import _lib as L
from .nerf_helpers import frequency_bands_cpu


PRECISIONS = {"fp32": L.PRECISION_FP32, "f16x3": L.PRECISION_F16X3, "f16x3_fwd": L.PRECISION_F16X3_FWD,
              "f16x3_fwd_dgrad": L.PRECISION_F16X3_FWD_DGRAD, "f16x3_train": L.PRECISION_F16X3_TRAIN}
TRAINING_PRECISIONS = ("fp32", "f16x3_fwd", "f16x3_fwd_dgrad", "f16x3_train")
INFERENCE_PRECISIONS = ("fp32", "f16x3")


class _PlanHandle:
    """Owner of one native ``nerfhip_plan`` (host-only object: packing tables, kernel schedules).  The handle is never
    duplicated: copying or unpickling an owner builds a NEW plan from the model configuration, so ``copy.deepcopy(model)``
    (EMA / best-model snapshots) and ``torch.save(model)`` cannot double-free or revive a stale address."""

    def __init__(self, cfg, precision=0):
        self.cfg = dict(cfg)
        self.precision = int(precision)
        lib = L.get_lib()
        mc = L.ModelCfg(**{k: int(v) for k, v in self.cfg.items()})
        self.ptr = lib.plan_create_ex(C.byref(mc), self.precision) if self.precision else lib.plan_create(C.byref(mc))
        if not self.ptr:
            raise L.NerfHipError("unsupported FlexibleNeRFModel geometry: " + lib.last_error().decode())
        fx = torch.zeros(16)
        fd = torch.zeros(16)
        nx, nd = self.cfg["num_encoding_fn_xyz"], self.cfg["num_encoding_fn_dir"]
        fx[:nx] = frequency_bands_cpu(nx, self.cfg["log_sampling_xyz"])
        if self.cfg["use_viewdirs"]:
            fd[:nd] = frequency_bands_cpu(nd, self.cfg["log_sampling_dir"])
        lib.plan_set_freqs(self.ptr, fx.contiguous().data_ptr(), fd.contiguous().data_ptr())

    def __del__(self):
        try:
            if getattr(self, "ptr", None):
                L.get_lib().plan_destroy(self.ptr)
                self.ptr = None
        except Exception:
            pass

    def __deepcopy__(self, memo):
        return _PlanHandle(self.cfg, self.precision)

    def __copy__(self):
        return _PlanHandle(self.cfg, self.precision)

    def __reduce__(self):
        return (_PlanHandle, (self.cfg, self.precision))


class _MlpFunction(torch.autograd.Function):
    """y = FlexibleNeRFModel(x).  Gradients flow to the parameters and -- when x requires grad (never in the render path:
    x is an encoding of constants) -- to x.  The parameters are passed as individual autograd inputs (they alias the model's flat buffer, which is what the
    kernels read); backward hands each one its slice of the flat gradient -- no concatenation in either direction."""

    @staticmethod
    def forward(ctx, model, x, need, need_x, *params):
        lib = L.get_lib()
        m = x.shape[0]
        out = torch.empty((m, 4), dtype=torch.float32, device=x.device)
        # `need` (decided by the caller: needs_input_grad ignores torch.no_grad()): keep the stash for a backward
        # (no backward will follow: the model's inference plan -- the fp32 one unless set_inference_precision chose otherwise)
        plan = model._plan if need else model._inference_plan()
        packed = model._packed() if need else model._inference_packed()
        stash = None
        if need:
            stash = torch.empty(max(lib.plan_stash_bytes(model._plan, m), 4) // 4, dtype=torch.float32, device=x.device)
        with L.launch_on(x, out, packed, stash) as st:
            lib.mlp_fwd(plan, packed.data_ptr(), x.data_ptr(), m, out.data_ptr(),
                        stash.data_ptr() if stash is not None else None, st)
        ctx.model, ctx.m, ctx.stash, ctx.packed = model, m, stash, packed
        # (the input gradient multiplies by the weights of this forward: keep a copy only if it will be asked for)
        ctx.flat = model._flat.clone() if (need and need_x) else None
        return out

    @staticmethod
    def backward(ctx, g_out):
        lib = L.get_lib()
        model, m = ctx.model, ctx.m
        g = g_out.contiguous().float()
        sb = lib.plan_bwd_scratch_bytes(model._plan, m)
        scratch = torch.empty(sb // 4 + 1, dtype=torch.float32, device=g.device)
        gflat = torch.empty(model.num_flat_params, dtype=torch.float32, device=g.device)
        with L.launch_on(g, scratch, gflat, ctx.packed, ctx.stash) as st:
            lib.mlp_bwd(model._plan, ctx.packed.data_ptr(), g.data_ptr(), m, ctx.stash.data_ptr(), scratch.data_ptr(), sb,
                        gflat.data_ptr(), st)
        gx = None
        if ctx.flat is not None and ctx.needs_input_grad[1]:
            gx = torch.empty((m, model.dim_xyz + model.dim_dir), dtype=torch.float32, device=g.device)
            with L.launch_on(scratch, ctx.flat, gx) as st:
                lib.mlp_bwd_input(model._plan, ctx.flat.data_ptr(), m, scratch.data_ptr(), gx.data_ptr(), st)
        return (None, gx, None, None) + model._split_flat(gflat)


class FlexibleNeRFModel(torch.nn.Module):
    def __init__(self, num_layers=4, hidden_size=128, skip_connect_every=4, num_encoding_fn_xyz=6,
                 num_encoding_fn_dir=4, include_input_xyz=True, include_input_dir=True, use_viewdirs=True,
                 log_sampling_xyz=True, log_sampling_dir=True):
        super().__init__()
        self.cfg = dict(num_layers=num_layers, hidden_size=hidden_size, skip_connect_every=skip_connect_every,
                        num_encoding_fn_xyz=num_encoding_fn_xyz, num_encoding_fn_dir=num_encoding_fn_dir,
                        include_input_xyz=bool(include_input_xyz), include_input_dir=bool(include_input_dir),
                        log_sampling_xyz=bool(log_sampling_xyz), log_sampling_dir=bool(log_sampling_dir),
                        use_viewdirs=bool(use_viewdirs))
        inc_xyz = 3 if include_input_xyz else 0
        inc_dir = 3 if include_input_dir else 0
        self.dim_xyz = inc_xyz + 2 * 3 * num_encoding_fn_xyz
        self.dim_dir = inc_dir + 2 * 3 * num_encoding_fn_dir
        self.skip_connect_every = skip_connect_every
        if not use_viewdirs:
            self.dim_dir = 0
        # Same construction order as the reference (models.py:205-229), so torch.manual_seed gives identical init.
        self.layer1 = torch.nn.Linear(self.dim_xyz, hidden_size)
        self.layers_xyz = torch.nn.ModuleList()
        for i in range(num_layers - 1):
            if i % self.skip_connect_every == 0 and i > 0 and i != num_layers - 1:
                self.layers_xyz.append(torch.nn.Linear(self.dim_xyz + hidden_size, hidden_size))
            else:
                self.layers_xyz.append(torch.nn.Linear(hidden_size, hidden_size))
        self.use_viewdirs = use_viewdirs
        if self.use_viewdirs:
            self.layers_dir = torch.nn.ModuleList()
            self.layers_dir.append(torch.nn.Linear(self.dim_dir + hidden_size, hidden_size // 2))
            self.fc_alpha = torch.nn.Linear(hidden_size, 1)
            self.fc_rgb = torch.nn.Linear(hidden_size // 2, 3)
            self.fc_feat = torch.nn.Linear(hidden_size, hidden_size)
        else:
            self.fc_out = torch.nn.Linear(hidden_size, 4)
        self.relu = torch.nn.functional.relu
        self._native_init()

    # ---- the native plan and the flat parameter storage ---------------------------------------------------------------
    def _native_init(self):
        """(Re)creates everything that refers to native memory: the plan, the tensor layout, the flat buffer."""
        lib = L.get_lib()
        self._plan_owner = _PlanHandle(self.cfg, PRECISIONS[getattr(self, "training_precision", "fp32")])
        self.num_flat_params = int(lib.plan_num_params(self._plan))
        self._layout = []
        for i in range(lib.plan_num_tensors(self._plan)):
            name, off, rows, cols = C.c_char_p(), C.c_int64(), C.c_int(), C.c_int()
            lib.plan_tensor_info(self._plan, i, C.byref(name), C.byref(off), C.byref(rows), C.byref(cols))
            self._layout.append((name.value.decode(), off.value, rows.value, cols.value))
        self._flat = None
        self._flat_grad = None
        self._pack_table = None
        self._packed_buf = None
        self._inf_owner = None
        self._inf_table = None
        self._inf_packed = None
        if getattr(self, "inference_precision", "fp32") != "fp32":
            self._inf_owner = _PlanHandle(self.cfg, PRECISIONS[self.inference_precision])
        # the backward's data flow: what set_backward_compaction last asked for; by default the fused one-kernel backward where the plan
        # has it (fp32 nets of hidden_size <= 64 with view directions, <= 4 layers, no skip layer: csrc/mlp64r.hip) -- over the
        # register-image stash (5; measured on MI355X, fern workload: 1.43 -> 1.15 ms per step against the recomputing mode 3) --, else dense
        if getattr(self, "_backward_choice", None) is None:
            self._fused_ok = None
            self.backward_compaction = self.fused_backward_available() or 0
        lib.plan_set_bwd_compaction(self._plan, int(self.backward_compaction))
        self._flatten()

    @property
    def _plan(self):
        return self._plan_owner.ptr

    # ---- arithmetic of the forward passes ------------------------------------------------------------------------------
    inference_precision = "fp32"
    training_precision = "fp32"

    def set_training_precision(self, precision):
        """Arithmetic of this model's TRAINING passes (and of its inference passes unless set_inference_precision says
        otherwise): "fp32" (default: the reference's own arithmetic, what the headline benchmark runs) or one of the
        fp16-piece plans of include/nerfhip.h -- three fp16 MFMAs per product on two IEEE fp16 pieces per operand, ~3 x 2^-24 per
        product, fp32-grade: "f16x3_fwd" (the forward) / "f16x3_fwd_dgrad" (+ the data-gradient chain) / "f16x3_train" (+ the large
        weight-gradient blocks).
        Parameters (re-homed into a fresh flat buffer, same Parameter objects), optimizer state and checkpoints are
        unaffected; call it before a TrainEngine is built on the model.  The two nets of a render may use different
        precisions (plans are per net)."""
        if precision not in TRAINING_PRECISIONS:
            raise ValueError("training precision must be one of %s (got %r)" % (TRAINING_PRECISIONS, precision))
        _PlanHandle(self.cfg, PRECISIONS[precision])  # (raises for a geometry the split-precision kernels do not cover, before anything changes)
        self.training_precision = precision
        self._native_init()
        return self

    backward_compaction = 0

    def set_backward_compaction(self, on=True):
        """Compacted backward (nerfhip_plan_set_bwd_compaction; off by default): this model's backward passes drop the sample points
        whose d(loss)/d(raw) row is exactly zero -- sigma_a = relu(raw[..., 3] + noise) is off there, or the ray's transmittance has
        reached 0 (nerf/volume_rendering_utils.py:38-42) -- instead of multiplying zeros through eight layers as autograd does
        (train_nerf.py:259).  The gradient is the same sum with its zero terms dropped; forward, stash, parameters and optimizer state
        are unaffected.  Works with every training precision; may be switched between steps.
        on = "recompute" (nerfhip_plan_set_bwd_compaction(plan, 2)): inside the fused render (run_one_iter_of_nerf / TrainEngine) the
        training forward additionally writes no activation stash; the backward re-runs the forward for the samples it keeps.  Pays
        where most rows are dropped and the stash-writing forward is much slower than the plain one (the fp16-piece plans).
        on = "fused" / "fused_compact" (3 / 4; fp32 nets of hidden_size <= 64 with view directions, <= 4 layers, no skip layer --
        config/fern.yml, config/llff.yml; raises for other geometries; "fused" is those nets' DEFAULT, on = False gives them the
        three-kernel dense backward): inside the fused render the forward writes no stash and ONE
        persistent kernel with the whole net resident in LDS recomputes the forward, runs the data-gradient chain and sums the weight
        gradients (csrc/mlp64r.hip) -- over every sample, or over the samples with a non-zero d(loss)/d(raw) row.
        on = "fused_stash" (5; same nets): the training forward leaves the chain's registers (encodings, every layer's activations)
        in a register-image stash -- whole-KiB stores, 1.8 KB per sample point -- and the same kernel reads them back instead of
        recomputing the forward: bit-identical gradient, a third fewer MFMAs in the backward, 2 x 1.8 KB of HBM traffic per sample.
        on = "auto": inside the fused render of the reference's own loop (run_one_iter_of_nerf -> loss.backward(), train_nerf.py:226-259)
        the mode of every training forward is chosen from the zero-row fraction this model's last compacted backward reported -- what
        TrainEngine(backward="auto") does for the engine's step: dense (64-wide nets: fused over the stash) while too little is dropped
        to pay for the list, compacted / recomputed / fused over the list from there on; a dense net is probed with one compacted
        pass every 50 forwards.  The two statistics words come back by an asynchronous copy that is polled, never waited for."""
        self._backward_choice = on
        if on == "auto":
            self._auto_frac, self._auto_calls, self._auto_event, self._auto_host = None, 0, None, None
            self.backward_compaction = self.fused_backward_available() or 0
        else:
            self.backward_compaction = {"recompute": 2, "fused": 3, "fused_compact": 4, "fused_stash": 5}.get(on, int(bool(on)))
        L.get_lib().plan_set_bwd_compaction(self._plan, self.backward_compaction)
        return self

    def _auto_choose_backward(self):
        """set_backward_compaction("auto"): sets the plan's mode for the training forward about to run (train_utils._FusedRender)."""
        if getattr(self, "_backward_choice", None) != "auto":
            return
        from .engine import TrainEngine
        if self._auto_event is not None and self._auto_event.query():
            kept, total = self._auto_host.tolist()
            if total > 0:
                self._auto_frac = 1.0 - kept / float(total)
            self._auto_event = None
        mode = TrainEngine._mode_for(self._auto_frac, self.training_precision != "fp32", self.fused_backward_available())
        probe = self._auto_calls % 50 == 0
        self._auto_calls += 1
        if probe and mode == 0:
            mode = 1
        if probe and mode in (3, 5):
            mode = 4
        if mode != self.backward_compaction:
            self.backward_compaction = mode
            L.get_lib().plan_set_bwd_compaction(self._plan, mode)

    def _auto_note_stats(self, words):
        """... and, behind a backward that ran over the list, asks for its {kept, total} (words: two int32 on the device; the copy is
        ordered behind the backward on the current stream and lands in pinned memory)."""
        if getattr(self, "_backward_choice", None) != "auto" or self._auto_event is not None:
            return
        if self._auto_host is None:
            self._auto_host = torch.zeros(2, dtype=torch.int32).pin_memory()
        self._auto_host.copy_(words, non_blocking=True)
        self._auto_event = torch.cuda.Event()
        self._auto_event.record(torch.cuda.current_stream(words.device))

    def fused_backward_available(self):
        """Truthy where set_backward_compaction("fused") works -- the plan has an LDS-resident image (csrc/nh_r64.h nh_r64_eligible):
        5 where "fused_stash" works too (the register-image stash fits the plan's stash region: every such plan today), else 3; 0
        where there is no fused backward."""
        cached = getattr(self, "_fused_ok", None)
        if cached is not None and cached[0] is self._plan:
            return cached[1]
        lib = L.get_lib()
        cur = lib.plan_bwd_compaction(self._plan)
        ok = 0
        for mode in (5, 3):
            try:
                lib.plan_set_bwd_compaction(self._plan, mode)
                ok = mode
                break
            except L.NerfHipError:
                pass
        lib.plan_set_bwd_compaction(self._plan, cur)
        self._fused_ok = (self._plan, ok)
        return ok

    def set_inference_precision(self, precision):
        """Arithmetic of this model's forward passes that no backward follows (torch.no_grad() / mode="validation"):
        "fp32" (default) or "f16x3" (NERFHIP_PRECISION_F16X3: fp16 pieces, fp32-grade products, > 2x the inference
        throughput).  Inference-only plans: the training precisions are
        set with set_training_precision.  Raises for geometries the split-precision kernels do not cover."""
        if precision not in INFERENCE_PRECISIONS:
            raise ValueError("inference precision must be one of %s (got %r)" % (INFERENCE_PRECISIONS, precision))
        owner = _PlanHandle(self.cfg, PRECISIONS[precision]) if precision != "fp32" else None
        self.inference_precision = precision
        self._inf_owner, self._inf_table, self._inf_packed = owner, None, None
        return self

    def _inference_plan(self):
        return self._plan if self._inf_owner is None else self._inf_owner.ptr

    def _inference_packed(self):
        if self._inf_owner is None:
            return self._packed()
        lib = L.get_lib()
        dev = self._flat.device
        if dev.type != "cuda":
            raise RuntimeError("FlexibleNeRFModel must live on a CUDA (HIP) device: nerf_pytorch_amd has no CPU path")
        plan = self._inf_owner.ptr
        if self._inf_table is None or self._inf_table.device != dev:
            n = int(lib.plan_packed_floats(plan))
            host = torch.empty(n, dtype=torch.int32)
            lib.plan_pack_index(plan, host.data_ptr())
            self._inf_table = host.to(dev)
            self._inf_packed = torch.empty(n, dtype=torch.float32, device=dev)
        # (re-packed on every call, like _packed(): ~30 us against a render chunk's milliseconds.  The parameters are written
        # behind torch's back -- the fused Adam kernel, views re-homed with .data -- so no version counter can be trusted)
        with L.launch_on(self._flat, self._inf_table, self._inf_packed) as st:
            lib.pack_weights_plan(plan, self._flat.data_ptr(), self._inf_table.data_ptr(), self._inf_packed.data_ptr(), st)
        return self._inf_packed

    def __getstate__(self):
        # copy.deepcopy / pickle: the native handle and every cache derived from it stay behind; the parameters travel
        # as ordinary tensors and are re-homed into a fresh flat buffer by __setstate__
        state = self.__dict__.copy()
        for k in ("_plan_owner", "_flat", "_flat_grad", "_pack_table", "_packed_buf", "_inf_owner", "_inf_table", "_inf_packed"):
            state[k] = None
        for k in ("_auto_event", "_auto_host"):   # (set_backward_compaction("auto"): the copy in flight stays behind)
            if k in state:
                state[k] = None
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        self._native_init()

    def _named(self):
        return dict(self.named_parameters())

    def _flatten(self):
        """Re-home every parameter as a view of one contiguous buffer, in state_dict order."""
        named = self._named()
        dev = next(iter(named.values())).device
        flat = torch.empty(self.num_flat_params, dtype=torch.float32, device=dev)
        for name, off, rows, cols in self._layout:
            p = named[name]
            n = rows * max(cols, 1)
            assert p.numel() == n, "parameter layout mismatch for %s" % name
            flat[off:off + n].copy_(p.detach().reshape(-1))
            p.data = flat[off:off + n].view(p.shape)
        self._flat = flat
        self._flat_grad = None
        self._pack_table = None
        self._packed_buf = None
        self._inf_table = None
        self._inf_packed = None

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._flatten()
        return r

    @property
    def flat_params(self):
        """The flat fp32 parameter vector every parameter aliases (reference state_dict order)."""
        return self._flat

    def flat_grad(self, attach=True):
        """A flat gradient vector; with attach=True every parameter's .grad becomes a view into it."""
        if self._flat_grad is None or self._flat_grad.device != self._flat.device:
            self._flat_grad = torch.zeros_like(self._flat)
        if attach:
            named = self._named()
            for name, off, rows, cols in self._layout:
                p = named[name]
                n = rows * max(cols, 1)
                p.grad = self._flat_grad[off:off + n].view(p.shape)
        return self._flat_grad

    def _packed(self, force=True):
        """MFMA-packed weight image of the current parameters (a ~10 us gather kernel)."""
        lib = L.get_lib()
        dev = self._flat.device
        if dev.type != "cuda":
            raise RuntimeError("FlexibleNeRFModel must live on a CUDA (HIP) device: nerf_pytorch_amd has no CPU path")
        n = int(lib.plan_packed_floats(self._plan))
        if self._pack_table is None or self._pack_table.device != dev:
            host = torch.empty(n, dtype=torch.int32)
            lib.plan_pack_index(self._plan, host.data_ptr())
            self._pack_table = host.to(dev)
            self._packed_buf = torch.empty(n, dtype=torch.float32, device=dev)
            force = True
        if force:
            with L.launch_on(self._flat, self._pack_table, self._packed_buf) as st:
                if self._plan_owner.precision:
                    lib.pack_weights_plan(self._plan, self._flat.data_ptr(), self._pack_table.data_ptr(), self._packed_buf.data_ptr(), st)
                else:
                    lib.pack_weights(self._flat.data_ptr(), self._pack_table.data_ptr(), n, self._packed_buf.data_ptr(), st)
        return self._packed_buf

    def _ordered_params(self):
        named = self._named()
        return [named[name] for name, _, _, _ in self._layout]

    def _split_flat(self, gflat):
        """Slices of a flat gradient vector shaped like the parameters, in _ordered_params() order."""
        return tuple(gflat[off:off + rows * max(cols, 1)].view((rows, cols) if cols else (rows,))
                     for _, off, rows, cols in self._layout)

    # ---- reference forward contract -----------------------------------------------------------------------------------
    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("FlexibleNeRFModel.forward needs CUDA (HIP) tensors: nerf_pytorch_amd has no CPU path")
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1]).contiguous().float()
        if x2.shape[-1] != self.dim_xyz + self.dim_dir:
            raise RuntimeError("expected %d input columns, got %d" % (self.dim_xyz + self.dim_dir, x2.shape[-1]))
        params = self._ordered_params()
        need_x = torch.is_grad_enabled() and x2.requires_grad
        need = need_x or (torch.is_grad_enabled() and any(p.requires_grad for p in params))
        y = _MlpFunction.apply(self, x2, need, need_x, *params)
        return y.reshape(list(lead) + [4])
