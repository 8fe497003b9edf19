"""ctypes binding of the C ABI declared in include/nerfhip.h.

``get_lib()`` loads the product library ``libnerfhip.so`` (hipcc, gfx950) that sits next to this file and fails loudly
when it is missing -- there is no CPU fallback in this package.  ``bind(path)`` is the generic binder (the test-suite
uses it to load the CPU wave-emulator build of the same sources; the product never does).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (diagnostic scripts under scripts/ that need an instrumented HIP build assign this attribute before the first
# get_lib(); the product reads no environment variable)
LIB_PATH = os.path.join(_HERE, "libnerfhip.so")

c_f = C.c_void_p  # device (or, for the emulator, host) pointers are passed as integers
c_i64 = C.c_int64
c_u64 = C.c_uint64
c_u32 = C.c_uint32


class ModelCfg(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "num_layers", "hidden_size", "skip_connect_every", "num_encoding_fn_xyz", "num_encoding_fn_dir",
        "include_input_xyz", "include_input_dir", "log_sampling_xyz", "log_sampling_dir", "use_viewdirs")]


class RenderCfg(C.Structure):
    _fields_ = [("num_coarse", C.c_int), ("num_fine", C.c_int), ("perturb", C.c_int), ("lindisp", C.c_int),
                ("white_background", C.c_int), ("noise_std", C.c_float), ("ray_stride", C.c_int)]


class RenderRand(C.Structure):
    _fields_ = [("t_rand", c_f), ("noise_coarse", c_f), ("u", c_f), ("noise_fine", c_f)]


class RenderOut(C.Structure):
    _fields_ = [(n, c_f) for n in ("rgb_coarse", "disp_coarse", "acc_coarse", "depth_coarse", "rgb_fine", "disp_fine",
                                   "acc_fine", "depth_fine")]


class RenderCotangents(C.Structure):
    _fields_ = [(n, c_f) for n in ("g_rgb_coarse", "g_acc_coarse", "g_depth_coarse", "g_rgb_fine", "g_acc_fine",
                                   "g_depth_fine")]


PRECISION_FP32 = 0  # NERFHIP_PRECISION_* (1 .. 4: round 3's bf16-piece plans, removed)
PRECISION_F16X3, PRECISION_F16X3_FWD, PRECISION_F16X3_FWD_DGRAD, PRECISION_F16X3_TRAIN = 5, 6, 7, 8
PART_COARSE, PART_FINE, PART_SHARED_BWD = 1, 2, 4


class SelectCfg(C.Structure):
    _fields_ = [("height", C.c_int32), ("width", C.c_int32), ("focal", C.c_float), ("near", C.c_float),
                ("far", C.c_float), ("use_viewdirs", C.c_int32), ("ndc", C.c_int32), ("ndc_near", C.c_float),
                ("ndc_cw", C.c_float), ("ndc_ch", C.c_float), ("ndc_two_near", C.c_float),
                ("ndc_neg_two_near", C.c_float), ("channels", C.c_int32), ("seed", c_u64), ("step", c_u64),
                ("first", c_i64)]


_PROTOS = {
    "nerfhip_version": (C.c_int, []),
    "nerfhip_last_error": (C.c_char_p, []),
    "nerfhip_is_emulated": (C.c_int, []),
    "nerfhip_profile_enable": (C.c_int, [C.c_int]),
    "nerfhip_profile_report": (C.c_int, [C.c_char_p, c_i64]),
    "nerfhip_profile_reserve": (C.c_int, [c_i64]),
    "nerfhip_profile_clocks": (C.c_int, [C.POINTER(c_u64)]),
    "nerfhip_rng_fill": (C.c_int, [C.c_int, c_u64, c_u32, c_u64, c_i64, c_f, c_f]),
    "nerfhip_ray_bundle": (C.c_int, [C.c_int, C.c_int, C.c_float, c_f, C.c_int, c_f, c_i64, c_f, c_f, c_f]),
    "nerfhip_ndc_rays": (C.c_int, [C.c_float] * 5 + [c_f, c_f, c_i64, c_f, c_f, c_f]),
    "nerfhip_plan_create_ex": (C.c_void_p, [C.c_void_p, C.c_int]),
    "nerfhip_plan_precision": (C.c_int, [C.c_void_p]),
    "nerfhip_pack_weights_plan": (C.c_int, [C.c_void_p, c_f, c_f, c_f, c_f]),
    "nerfhip_ndc_rays_bwd": (C.c_int, [C.c_float] * 5 + [c_f, c_f, c_f, c_f, c_i64, c_f, c_f, c_f]),
    "nerfhip_pack_rays": (C.c_int, [c_f, c_f, c_f, C.c_float, C.c_float, c_i64, c_f, c_f]),
    "nerfhip_positional_encoding": (C.c_int, [c_f, c_i64, C.c_int, c_f, C.c_int, C.c_int, c_f, c_f]),
    "nerfhip_stratified_z": (C.c_int, [c_f, C.c_int, c_i64, c_f, C.c_int, C.c_int, C.c_int, c_f, c_u64, c_u64, c_f,
                                        c_f]),
    "nerfhip_cumprod_exclusive": (C.c_int, [c_f, c_i64, C.c_int, c_f, c_f]),
    "nerfhip_cumprod_exclusive_bwd": (C.c_int, [c_f, c_f, c_f, c_i64, C.c_int, c_f, c_f]),
    "nerfhip_volume_render_fwd": (C.c_int, [c_f, c_f, c_f, C.c_int, c_i64, C.c_int, C.c_float, c_f, c_u64, c_u32, c_u64,
                                             C.c_int, c_f, c_f, c_f, c_f, c_f, c_f]),
    "nerfhip_volume_render_bwd": (C.c_int, [c_f, c_f, c_f, C.c_int, c_i64, C.c_int, C.c_float, c_f, c_u64, c_u32, c_u64,
                                             C.c_int, c_f, c_f, c_f, c_f, c_f, c_f]),
    "nerfhip_sample_pdf": (C.c_int, [c_f, c_f, c_i64, C.c_int, c_f, C.c_int, c_f, C.c_int, c_u64, c_u64, c_f, c_f, c_f,
                                      c_f]),
    "nerfhip_hierarchical_z": (C.c_int, [c_f, c_f, c_i64, C.c_int, c_f, C.c_int, c_f, C.c_int, c_u64, c_u64, c_f, c_f,
                                          c_f]),
    "nerfhip_plan_create": (C.c_void_p, [C.POINTER(ModelCfg)]),
    "nerfhip_plan_destroy": (None, [C.c_void_p]),
    "nerfhip_plan_num_params": (c_i64, [C.c_void_p]),
    "nerfhip_plan_dim_xyz": (C.c_int, [C.c_void_p]),
    "nerfhip_plan_dim_dir": (C.c_int, [C.c_void_p]),
    "nerfhip_plan_num_tensors": (C.c_int, [C.c_void_p]),
    "nerfhip_plan_tensor_info": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(c_i64),
                                            C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "nerfhip_plan_packed_floats": (c_i64, [C.c_void_p]),
    "nerfhip_plan_describe": (C.c_int, [C.c_void_p, C.c_char_p, c_i64]),
    "nerfhip_plan_pack_index": (C.c_int, [C.c_void_p, C.c_void_p]),
    "nerfhip_pack_weights": (C.c_int, [c_f, c_f, c_i64, c_f, c_f]),
    "nerfhip_plan_stash_bytes": (c_i64, [C.c_void_p, c_i64]),
    "nerfhip_plan_bwd_scratch_bytes": (c_i64, [C.c_void_p, c_i64]),
    "nerfhip_plan_set_freqs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "nerfhip_plan_set_bwd_compaction": (C.c_int, [C.c_void_p, C.c_int]),
    "nerfhip_plan_bwd_compaction": (C.c_int, [C.c_void_p]),
    "nerfhip_plan_bwd_stats_offset": (c_i64, [C.c_void_p, c_i64]),
    "nerfhip_mlp_fwd": (C.c_int, [C.c_void_p, c_f, c_f, c_i64, c_f, c_f, c_f]),
    "nerfhip_mlp_bwd": (C.c_int, [C.c_void_p, c_f, c_f, c_i64, c_f, c_f, c_i64, c_f, c_f]),
    "nerfhip_mlp_bwd_input": (C.c_int, [C.c_void_p, c_f, c_i64, c_f, c_f, c_f]),
    "nerfhip_render_workspace_bytes": (c_i64, [C.c_void_p, C.c_void_p, C.POINTER(RenderCfg), c_i64, C.c_int]),
    "nerfhip_render_workspace_region": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(RenderCfg), c_i64, C.c_int, C.c_char_p,
                                                   C.POINTER(c_i64), C.POINTER(c_i64)]),
    "nerfhip_render_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(RenderCfg), c_f, c_i64, c_f, c_f, c_f, c_f,
                                      C.POINTER(RenderRand), c_u64, c_u64, C.POINTER(RenderOut), c_f, c_i64, C.c_int,
                                      c_f]),
    "nerfhip_render_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(RenderCfg), c_f, c_i64, c_f, c_f,
                                      C.POINTER(RenderRand), c_u64, c_u64, c_f, c_f, c_f, c_i64, c_f, c_f, c_f]),
    "nerfhip_render_fwd_parts": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(RenderCfg), c_f, c_i64, c_f, c_f, c_f, c_f,
                                            C.POINTER(RenderRand), c_u64, c_u64, C.POINTER(RenderOut), c_f, c_i64,
                                            C.c_int, C.c_int, c_f]),
    "nerfhip_render_bwd_parts": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(RenderCfg), c_f, c_i64, c_f, c_f,
                                            C.POINTER(RenderRand), c_u64, c_u64, C.POINTER(RenderCotangents), c_f, c_i64,
                                            c_f, c_f, C.c_int, c_f]),
    "nerfhip_render_bwd_rays_tmp_bytes": (c_i64, [C.c_void_p, C.c_void_p, C.POINTER(RenderCfg), c_i64]),
    "nerfhip_render_bwd_rays": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(RenderCfg), c_f, c_i64, c_f, c_f,
                                           C.POINTER(RenderRand), c_u64, c_u64, C.POINTER(RenderCotangents), c_f, c_i64,
                                           c_f, c_f, C.c_int, c_f, c_f, c_f, c_i64, c_f, c_f]),
    "nerfhip_mse_loss_fwd_bwd": (C.c_int, [c_f, c_f, c_f, C.c_int, c_i64, C.c_float, c_f, c_f, c_f, c_f]),
    "nerfhip_adam_step": (C.c_int, [c_f, c_f, c_f, c_f, c_i64, C.c_float, C.c_float, C.c_float, C.c_float, c_i64,
                                     C.c_float, c_f]),
    "nerfhip_select_indices": (C.c_int, [c_u64, c_u64, c_i64, c_i64, c_i64, c_f, c_f]),
    "nerfhip_select_rays": (C.c_int, [C.POINTER(SelectCfg), c_f, C.c_int, c_f, c_f, c_i64, c_f, c_f, c_f, c_f]),
    "nerfhip_select_cached_rays": (C.c_int, [C.POINTER(SelectCfg), c_f, c_f, c_f, c_i64, c_f, c_i64, c_f, c_f, c_f,
                                             c_f]),
    "nerfhip_cast_to_image": (C.c_int, [c_f, C.c_int, c_i64, c_f, c_f]),
    "nerfhip_cast_to_disparity_image": (C.c_int, [c_f, c_i64, c_f, c_f, c_f]),
}

EXPORTED_SYMBOLS = tuple(sorted(_PROTOS))


class NerfHipError(RuntimeError):
    pass


class NerfHipLib:
    """A loaded libnerfhip with typed prototypes; every int-returning entry point raises NerfHipError on failure."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise NerfHipError(
                "%s not found: build it with `python __graft_entry__.py build` (hipcc --offload-arch=gfx950). "
                "This package has no CPU fallback." % path)
        self.path = path
        self._dll = C.CDLL(path)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(self._dll, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
            if res is C.c_int and name not in ("nerfhip_version", "nerfhip_is_emulated", "nerfhip_plan_dim_xyz",
                                              "nerfhip_plan_dim_dir", "nerfhip_plan_num_tensors", "nerfhip_plan_precision",
                                              "nerfhip_plan_bwd_compaction"):
                setattr(self, name[len("nerfhip_"):], self._checked(fn, name))
            else:
                setattr(self, name[len("nerfhip_"):], fn)

    def _checked(self, fn, name):
        def call(*a):
            rc = fn(*a)
            if rc != 0:
                msg = self._dll.nerfhip_last_error()
                raise NerfHipError("%s failed (%d): %s" % (name, rc, msg.decode() if msg else ""))
            return rc
        call.__name__ = name
        return call


def bind(path):
    return NerfHipLib(path)


class launch_on:
    """``with launch_on(t0, t1, ...) as stream:`` -- the tensors handed to a C-ABI call must all live on ONE CUDA (HIP)
    device; that device is made current for the duration of the launch and `stream` is ITS current stream (not the
    current stream of whatever device happens to be current in a process that drives several GPUs)."""

    def __init__(self, *tensors):
        import torch
        dev = None
        for t in tensors:
            if t is None:
                continue
            if not t.is_cuda:
                raise RuntimeError("nerf_pytorch_amd has no CPU path: got a %s tensor" % t.device)
            if dev is None:
                dev = t.device
            elif t.device != dev:
                raise RuntimeError("tensors of one call live on different devices (%s and %s)" % (dev, t.device))
        if dev is None:
            raise RuntimeError("launch_on needs at least one tensor")
        self.dev = dev
        self._guard = torch.cuda.device(dev)
        self._stream = torch.cuda.current_stream(dev).cuda_stream

    def __enter__(self):
        self._guard.__enter__()
        return self._stream

    def __exit__(self, *exc):
        return self._guard.__exit__(*exc)


_LIB = None
# `make variant` builds (A/B schedules, wrong-result cost-attribution switches: csrc/nh_diag.h) report nerfhip_version() + DIAG_FLAG.
# The package refuses them; a diagnostic script under scripts/ that loads one on purpose sets ALLOW_DIAG (and LIB_PATH) first.
DIAG_FLAG = 1000000
ALLOW_DIAG = False


def get_lib():
    """The product library.  Imported lazily so that `import nerf_pytorch_amd` works on a box without the .so, but any
    compute call fails loudly."""
    global _LIB
    if _LIB is None:
        import torch  # noqa: F401  -- import torch first so that the process uses torch's HIP runtime
        _LIB = NerfHipLib(LIB_PATH)
        if _LIB.is_emulated():
            raise NerfHipError("libnerfhip.so reports an emulator build; refusing to use it as the product path")
        if _LIB.version() >= DIAG_FLAG and not ALLOW_DIAG:
            lib, _LIB = _LIB, None
            raise NerfHipError("%s is a `make variant` build (A/B or diagnostic switches, csrc/nh_diag.h): not the product library" % lib.path)
    return _LIB
