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


def _stream():
    return torch.cuda.current_stream().cuda_stream


class _MlpFunction(torch.autograd.Function):
    """y = FlexibleNeRFModel(x).  Gradients flow to the parameters only (x is an encoding: no gradient in the hot path)."""

    @staticmethod
    def forward(ctx, model, x, flat):
        lib = L.get_lib()
        m = x.shape[0]
        out = torch.empty((m, 4), dtype=torch.float32, device=x.device)
        need = bool(ctx.needs_input_grad[2])  # (grad mode is off inside Function.forward: ask autograd instead)
        packed = model._packed()
        stash = None
        if need:
            stash = torch.empty(max(lib.plan_stash_bytes(model._plan, m), 4) // 4, dtype=torch.float32, device=x.device)
        lib.mlp_fwd(model._plan, packed.data_ptr(), x.data_ptr(), m, out.data_ptr(),
                    stash.data_ptr() if stash is not None else None, _stream())
        ctx.model, ctx.m, ctx.stash, ctx.packed = model, m, stash, packed
        return out

    @staticmethod
    def backward(ctx, g_out):
        lib = L.get_lib()
        model, m = ctx.model, ctx.m
        g = g_out.contiguous().float()
        sb = lib.plan_bwd_scratch_bytes(model._plan, m)
        scratch = torch.empty(sb // 4 + 1, dtype=torch.float32, device=g.device)
        gflat = torch.empty(model.num_flat_params, dtype=torch.float32, device=g.device)
        lib.mlp_bwd(model._plan, ctx.packed.data_ptr(), g.data_ptr(), m, ctx.stash.data_ptr(), scratch.data_ptr(), sb,
                    gflat.data_ptr(), _stream())
        return None, None, gflat


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

        lib = L.get_lib()
        mc = L.ModelCfg(**{k: int(v) for k, v in self.cfg.items()})
        self._plan = lib.plan_create(C.byref(mc))
        if not self._plan:
            raise L.NerfHipError("unsupported FlexibleNeRFModel geometry: " + lib.last_error().decode())
        fx = torch.zeros(16)
        fd = torch.zeros(16)
        fx[:num_encoding_fn_xyz] = frequency_bands_cpu(num_encoding_fn_xyz, log_sampling_xyz)
        if use_viewdirs:
            fd[:num_encoding_fn_dir] = frequency_bands_cpu(num_encoding_fn_dir, log_sampling_dir)
        self._freq_keep = (fx.contiguous(), fd.contiguous())
        lib.plan_set_freqs(self._plan, fx.data_ptr(), fd.data_ptr())
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
        self._packed_version = None
        self._flatten()

    def __del__(self):
        try:
            if getattr(self, "_plan", None):
                L.get_lib().plan_destroy(self._plan)
                self._plan = None
        except Exception:
            pass

    # ---- flat parameter storage -------------------------------------------------------------------------------------
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
            lib.pack_weights(self._flat.data_ptr(), self._pack_table.data_ptr(), n, self._packed_buf.data_ptr(), _stream())
        return self._packed_buf

    # ---- reference forward contract -----------------------------------------------------------------------------------
    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("FlexibleNeRFModel.forward needs CUDA (HIP) tensors: nerf_pytorch_amd has no CPU path")
        if x.requires_grad:
            raise RuntimeError("gradients w.r.t. the encoded input are not supported (the hot path never needs them)")
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1]).contiguous().float()
        if x2.shape[-1] != self.dim_xyz + self.dim_dir:
            raise RuntimeError("expected %d input columns, got %d" % (self.dim_xyz + self.dim_dir, x2.shape[-1]))
        # route the gradient to the parameters through the flat view: grads of views accumulate into each .grad
        flat_leaf = torch.cat([p.reshape(-1) for p in self._ordered_params()]) if torch.is_grad_enabled() and any(
            p.requires_grad for p in self.parameters()) else self._flat
        y = _MlpFunction.apply(self, x2, flat_leaf)
        return y.reshape(list(lead) + [4])

    def _ordered_params(self):
        named = self._named()
        return [named[name] for name, _, _, _ in self._layout]
