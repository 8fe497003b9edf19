"""Test harness: drives the C ABI (include/nerfhip.h) on numpy inputs through one of two backends.

* ``EmuBackend``  -- tests/emu/libnerfhip_emu.so: the kernel sources compiled for the CPU wave emulator.  Host
  pointers; used by the ``-m "not gpu"`` suite to check kernel index algebra without a GPU.  Test infrastructure.
* ``GpuBackend``  -- the product library nerf-pytorch_amd/libnerfhip.so on cuda:0 (torch owns the device memory).

Both expose the same numpy-in / numpy-out helpers, so every parity test is written once.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import nerf_pytorch_amd._lib as L  # noqa: E402

EMU_SO = os.path.join(ROOT, "tests", "emu", "libnerfhip_emu.so")
CSRC = os.path.join(ROOT, "nerf-pytorch_amd", "csrc")


def build_emu():
    subprocess.run(["make", "-C", CSRC, "emu", "-j8"], check=True, stdout=subprocess.DEVNULL)
    return EMU_SO


def model_cfg(num_layers=4, hidden_size=128, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4,
              include_input_xyz=True, include_input_dir=True, log_sampling_xyz=True, log_sampling_dir=True,
              use_viewdirs=True):
    return dict(num_layers=num_layers, hidden_size=hidden_size, skip_connect_every=skip_connect_every,
                num_encoding_fn_xyz=num_encoding_fn_xyz, num_encoding_fn_dir=num_encoding_fn_dir,
                include_input_xyz=include_input_xyz, include_input_dir=include_input_dir,
                log_sampling_xyz=log_sampling_xyz, log_sampling_dir=log_sampling_dir, use_viewdirs=use_viewdirs)


class Backend:
    name = "?"

    # -- device array plumbing (overridden) --
    def dev(self, a):
        raise NotImplementedError

    def empty(self, shape, dtype=np.float32):
        raise NotImplementedError

    def ptr(self, d):
        raise NotImplementedError

    def host(self, d):
        raise NotImplementedError

    def stream(self):
        return None

    def p(self, d):
        return None if d is None else self.ptr(d)

    def devopt(self, a, dtype=np.float32):
        return None if a is None else self.dev(np.ascontiguousarray(a, dtype=dtype))

    # -- rows either side of the path: ray selection, 8-bit output ---------------------------------------------------
    def select_indices(self, seed, step, population, first, n):
        out = self.empty((n,), np.int64)
        self.lib.select_indices(seed, step, population, first, n, self.ptr(out), self.stream())
        return self.host(out)

    def _select_cfg(self, H, W, focal, near, far, use_viewdirs, ndc, channels, seed, step, first):
        f32 = np.float32
        return L.SelectCfg(height=H, width=W, focal=float(focal), near=float(near), far=float(far),
                           use_viewdirs=int(use_viewdirs), ndc=int(ndc), ndc_near=1.0,
                           ndc_cw=float(f32(-1.0 / (W / (2.0 * focal)))), ndc_ch=float(f32(-1.0 / (H / (2.0 * focal)))),
                           ndc_two_near=2.0, ndc_neg_two_near=-2.0, channels=channels, seed=seed, step=step, first=first)

    def select_rays(self, H, W, focal, c2w, image, n, near, far, inds=None, use_viewdirs=True, ndc=False, seed=0, step=0,
                    first=0):
        c2w = np.ascontiguousarray(c2w, np.float32)
        ch = 3 if image is None else image.shape[-1]
        cfg = self._select_cfg(H, W, focal, near, far, use_viewdirs, ndc, ch, seed, step, first)
        dc, di, dn = self.dev(c2w), self.devopt(image), self.devopt(inds, np.int64)
        rays, tgt, used = self.empty((n, 11 if use_viewdirs else 8)), self.empty((n, ch)), self.empty((n,), np.int64)
        self.lib.select_rays(C.byref(cfg), self.ptr(dc), c2w.shape[1], self.p(di), self.p(dn), n, self.ptr(rays),
                             self.ptr(tgt) if image is not None else None, self.ptr(used), self.stream())
        return self.host(rays), (self.host(tgt) if image is not None else None), self.host(used)

    def select_cached_rays(self, H, W, focal, ro, rd, targets, n, near, far, inds=None, use_viewdirs=True, ndc=False,
                           seed=0, step=0, first=0):
        ch = targets.shape[-1]
        cfg = self._select_cfg(H, W, focal, near, far, use_viewdirs, ndc, ch, seed, step, first)
        do, dd, dt, dn = self.dev(ro), self.dev(rd), self.dev(targets), self.devopt(inds, np.int64)
        rays, tgt, used = self.empty((n, 11 if use_viewdirs else 8)), self.empty((n, ch)), self.empty((n,), np.int64)
        self.lib.select_cached_rays(C.byref(cfg), self.ptr(do), self.ptr(dd), self.ptr(dt), ro.shape[0], self.p(dn), n,
                                    self.ptr(rays), self.ptr(tgt), self.ptr(used), self.stream())
        return self.host(rays), self.host(tgt), self.host(used)

    def cast_to_image(self, rgb):
        h, w, c = rgb.shape
        d = self.dev(np.ascontiguousarray(rgb, np.float32))
        out = self.empty((h, w, 3), np.uint8)
        self.lib.cast_to_image(self.ptr(d), c, h * w, self.ptr(out), self.stream())
        return self.host(out)

    def cast_to_disparity_image(self, disp):
        d = self.dev(np.ascontiguousarray(disp, np.float32))
        out, scratch = self.empty(disp.shape, np.uint8), self.empty((3,))
        self.lib.cast_to_disparity_image(self.ptr(d), disp.size, self.ptr(scratch), self.ptr(out), self.stream())
        return self.host(out)

    # -- unit ops ---------------------------------------------------------------------------------------------------
    def rng_fill(self, kind, seed, stream_id, first, n):
        out = self.empty((n,))
        self.lib.rng_fill(kind, seed, stream_id, first, n, self.ptr(out), self.stream())
        return self.host(out)

    def ray_bundle(self, H, W, focal, c2w, pixels=None):
        c2w = np.ascontiguousarray(c2w, np.float32)
        ld = c2w.shape[1]
        n = H * W if pixels is None else len(pixels)
        dc, dp = self.dev(c2w), self.devopt(pixels, np.int64)
        ro, rd = self.empty((n, 3)), self.empty((n, 3))
        self.lib.ray_bundle(H, W, float(focal), self.ptr(dc), ld, self.p(dp), n, self.ptr(ro), self.ptr(rd), self.stream())
        return self.host(ro), self.host(rd)

    def ndc_rays(self, H, W, focal, near, ro, rd):
        cw = np.float32(-1.0 / (W / (2.0 * focal)))
        ch = np.float32(-1.0 / (H / (2.0 * focal)))
        n = ro.shape[0]
        dro, drd = self.dev(ro.astype(np.float32)), self.dev(rd.astype(np.float32))
        oo, od = self.empty((n, 3)), self.empty((n, 3))
        self.lib.ndc_rays(float(near), float(cw), float(ch), float(np.float32(2.0 * near)), float(np.float32(-2.0 * near)),
                          self.ptr(dro), self.ptr(drd), n, self.ptr(oo), self.ptr(od), self.stream())
        return self.host(oo), self.host(od)

    def ndc_rays_bwd(self, H, W, focal, near, ro, rd, g_oo, g_od):
        cw = np.float32(-1.0 / (W / (2.0 * focal)))
        ch = np.float32(-1.0 / (H / (2.0 * focal)))
        n = ro.shape[0]
        d = [self.dev(np.ascontiguousarray(a, np.float32)) for a in (ro, rd, g_oo, g_od)]
        g_ro, g_rd = self.empty((n, 3)), self.empty((n, 3))
        self.lib.ndc_rays_bwd(float(near), float(cw), float(ch), float(np.float32(2.0 * near)), float(np.float32(-2.0 * near)),
                              *[self.ptr(a) for a in d], n, self.ptr(g_ro), self.ptr(g_rd), self.stream())
        return self.host(g_ro), self.host(g_rd)

    def pack_rays(self, ro, rd, near, far, viewdir_src=None):
        n = ro.shape[0]
        dro, drd, dv = self.dev(ro), self.dev(rd), self.devopt(viewdir_src)
        out = self.empty((n, 11 if viewdir_src is not None else 8))
        self.lib.pack_rays(self.ptr(dro), self.ptr(drd), self.p(dv), float(near), float(far), n, self.ptr(out), self.stream())
        return self.host(out)

    def positional_encoding(self, x, freqs, include_input):
        m, d = x.shape
        nf = len(freqs)
        dx = self.dev(np.ascontiguousarray(x, np.float32))
        df = self.dev(np.ascontiguousarray(freqs, np.float32)) if nf else None
        out = self.empty((m, d * (int(include_input) + 2 * nf)))
        self.lib.positional_encoding(self.ptr(dx), m, d, self.p(df), nf, int(include_input), self.ptr(out), self.stream())
        return self.host(out)

    def stratified_z(self, rays, t_vals, lindisp, perturb, t_rand=None, seed=0, ray_offset=0):
        n, stride = rays.shape
        nc = len(t_vals)
        dr, dt, dtr = self.dev(rays), self.dev(np.ascontiguousarray(t_vals, np.float32)), self.devopt(t_rand)
        z = self.empty((n, nc))
        self.lib.stratified_z(self.ptr(dr), stride, n, self.ptr(dt), nc, int(lindisp), int(perturb), self.p(dtr), seed,
                              ray_offset, self.ptr(z), self.stream())
        return self.host(z)

    def cumprod_exclusive(self, x):
        rows, cols = x.shape
        dx = self.dev(np.ascontiguousarray(x, np.float32))
        out = self.empty((rows, cols))
        self.lib.cumprod_exclusive(self.ptr(dx), rows, cols, self.ptr(out), self.stream())
        return self.host(out)

    def volume_render_fwd(self, raw, z, rd, noise_std=0.0, noise=None, white=False, seed=0, rng_stream=1, ray_offset=0):
        n, s = z.shape
        draw, dz, drd, dn = self.dev(raw), self.dev(z), self.dev(np.ascontiguousarray(rd, np.float32)), self.devopt(noise)
        rgb, disp, acc, w, dep = self.empty((n, 3)), self.empty((n,)), self.empty((n,)), self.empty((n, s)), self.empty((n,))
        self.lib.volume_render_fwd(self.ptr(draw), self.ptr(dz), self.ptr(drd), rd.shape[1], n, s, float(noise_std),
                                   self.p(dn), seed, rng_stream, ray_offset, int(white), self.ptr(rgb), self.ptr(disp),
                                   self.ptr(acc), self.ptr(w), self.ptr(dep), self.stream())
        return tuple(self.host(t) for t in (rgb, disp, acc, w, dep))

    def volume_render_bwd(self, raw, z, rd, g_rgb=None, g_depth=None, g_acc=None, g_weights=None, noise_std=0.0,
                          noise=None, white=False, seed=0, rng_stream=1, ray_offset=0):
        n, s = z.shape
        draw, dz, drd, dn = self.dev(raw), self.dev(z), self.dev(np.ascontiguousarray(rd, np.float32)), self.devopt(noise)
        gs = [self.devopt(g) for g in (g_rgb, g_depth, g_acc, g_weights)]
        out = self.empty((n, s, 4))
        self.lib.volume_render_bwd(self.ptr(draw), self.ptr(dz), self.ptr(drd), rd.shape[1], n, s, float(noise_std),
                                   self.p(dn), seed, rng_stream, ray_offset, int(white), self.p(gs[0]), self.p(gs[1]),
                                   self.p(gs[2]), self.p(gs[3]), self.ptr(out), self.stream())
        return self.host(out)

    def sample_pdf(self, bins, weights, nf, u=None, det=False, seed=0, ray_offset=0):
        n, nb = bins.shape
        db, dw, du = self.dev(bins), self.dev(weights), self.devopt(u)
        dud = self.dev(self._linspace01(nf)) if det else None
        s, inds, cdf = self.empty((n, nf)), self.empty((n, nf), np.int64), self.empty((n, nb))
        self.lib.sample_pdf(self.ptr(db), self.ptr(dw), n, nb, self.p(du), int(det), self.p(dud), nf, seed, ray_offset,
                            self.ptr(s), self.ptr(inds), self.ptr(cdf), self.stream())
        return self.host(s), self.host(inds), self.host(cdf)

    def hierarchical_z(self, z, w, nf, u=None, det=False, seed=0, ray_offset=0):
        n, nc = z.shape
        dz, dw, du = self.dev(z), self.dev(w), self.devopt(u)
        dud = self.dev(self._linspace01(nf)) if det else None
        zs, zf = self.empty((n, nf)), self.empty((n, nc + nf))
        self.lib.hierarchical_z(self.ptr(dz), self.ptr(dw), n, nc, self.p(du), int(det), self.p(dud), nf, seed, ray_offset,
                                self.ptr(zs), self.ptr(zf), self.stream())
        return self.host(zs), self.host(zf)

    @staticmethod
    def _linspace01(n):
        import torch
        return torch.linspace(0.0, 1.0, n).numpy()

    # -- MLP ------------------------------------------------------------------------------------------------------------
    def make_plan(self, cfg, precision=0):
        import torch
        mc = L.ModelCfg(**{k: int(v) for k, v in cfg.items()})
        plan = self.lib.plan_create_ex(C.byref(mc), int(precision)) if precision else self.lib.plan_create(C.byref(mc))
        if not plan:
            raise L.NerfHipError(self.lib.last_error().decode())
        import nerf_oracle as O
        fx = np.zeros(16, np.float32)
        fd = np.zeros(16, np.float32)
        fx[:cfg["num_encoding_fn_xyz"]] = O.frequency_bands(cfg["num_encoding_fn_xyz"], cfg["log_sampling_xyz"]).numpy()
        if cfg["use_viewdirs"]:
            fd[:cfg["num_encoding_fn_dir"]] = O.frequency_bands(cfg["num_encoding_fn_dir"], cfg["log_sampling_dir"]).numpy()
        self.lib.plan_set_freqs(plan, fx.ctypes.data, fd.ctypes.data)
        return plan

    def tensor_table(self, plan):
        out = []
        for i in range(self.lib.plan_num_tensors(plan)):
            name, off, rows, cols = C.c_char_p(), C.c_int64(), C.c_int(), C.c_int()
            self.lib.plan_tensor_info(plan, i, C.byref(name), C.byref(off), C.byref(rows), C.byref(cols))
            out.append((name.value.decode(), off.value, rows.value, cols.value))
        return out

    def flatten_params(self, plan, params):
        flat = np.zeros(self.lib.plan_num_params(plan), np.float32)
        for name, off, rows, cols in self.tensor_table(plan):
            a = np.asarray(params[name], np.float32).reshape(-1)
            assert a.size == rows * max(cols, 1), name
            flat[off:off + a.size] = a
        return flat

    def unflatten(self, plan, flat):
        out = {}
        for name, off, rows, cols in self.tensor_table(plan):
            n = rows * max(cols, 1)
            out[name] = flat[off:off + n].reshape((rows, cols) if cols else (rows,))
        return out

    def pack(self, plan, flat):
        n = self.lib.plan_packed_floats(plan)
        table = np.empty(n, np.int32)
        self.lib.plan_pack_index(plan, table.ctypes.data)
        dt, dp = self.dev(table), self.dev(flat)
        packed = self.empty((n,))
        if self.lib.plan_precision(plan) != 0:  # (fp32 plans keep exercising the plan-independent entry point)
            self.lib.pack_weights_plan(plan, self.ptr(dp), self.ptr(dt), self.ptr(packed), self.stream())
        else:
            self.lib.pack_weights(self.ptr(dp), self.ptr(dt), n, self.ptr(packed), self.stream())
        return packed  # device array

    def mlp_fwd(self, plan, packed, x, want_stash=False):
        m = x.shape[0]
        dx = self.dev(np.ascontiguousarray(x, np.float32))
        out = self.empty((m, 4))
        stash = self.empty((max(self.lib.plan_stash_bytes(plan, m) // 4, 1),)) if want_stash else None
        self.lib.mlp_fwd(plan, self.ptr(packed), self.ptr(dx), m, self.ptr(out), self.p(stash), self.stream())
        return self.host(out), stash

    def set_compaction(self, plan, on):
        """nerfhip_plan_set_bwd_compaction: the plan's backward drops the samples whose d(raw output) row is all zero."""
        mode = {"recompute": 2, "fused": 3, "fused_compact": 4, "fused_stash": 5}.get(on, int(bool(on)))
        self.lib.plan_set_bwd_compaction(plan, mode)
        assert self.lib.plan_bwd_compaction(plan) == mode

    def bwd_stats(self, plan, m, scratch):
        """(samples kept, samples of the launch) of the last compacted backward that ran in `scratch`."""
        off = self.lib.plan_bwd_stats_offset(plan, m)
        assert off >= 0 and off % 4 == 0
        w = self.host(scratch[off // 4:off // 4 + 2])
        return tuple(int(v) for v in np.ascontiguousarray(w).view(np.int32))

    def mlp_bwd(self, plan, packed, g_out, stash, flat_for_input_grad=None, want_stats=False):
        """Flat parameter gradient; with `flat_for_input_grad` (the flat parameter vector) also d(loss)/d(x); with want_stats
        also the compacted backward's (kept, total) sample counts."""
        m = g_out.shape[0]
        dg = self.dev(np.ascontiguousarray(g_out, np.float32))
        sb = self.lib.plan_bwd_scratch_bytes(plan, m)
        scratch = self.empty((sb // 4,))
        gp = self.empty((self.lib.plan_num_params(plan),))
        self.lib.mlp_bwd(plan, self.ptr(packed), self.ptr(dg), m, self.ptr(stash), self.ptr(scratch), sb, self.ptr(gp),
                         self.stream())
        if want_stats:
            return self.host(gp), self.bwd_stats(plan, m, scratch)
        if flat_for_input_grad is None:
            return self.host(gp)
        d = self.lib.plan_dim_xyz(plan) + self.lib.plan_dim_dir(plan)
        gx = self.empty((m, d))
        dflat = self.dev(np.ascontiguousarray(flat_for_input_grad, np.float32))
        self.lib.mlp_bwd_input(plan, self.ptr(dflat), m, self.ptr(scratch), self.ptr(gx), self.stream())
        return self.host(gp), self.host(gx)

    # -- fused render ---------------------------------------------------------------------------------------------------
    def render(self, plan_c, plan_f, packed_c, packed_f, rays, opt, rand=None, seed=0, ray_offset=0, training=False,
               g_rgb=None, want_regions=(), ray_grad_params=None):
        """opt: dict(num_coarse, num_fine, perturb, lindisp, white_background, noise_std).  Returns dict of outputs
        (+ flat grads 'g_params_coarse/fine' when g_rgb = (g_c, g_f), or a callable that makes them from the outputs, is given)."""
        rand = rand or {}
        n, stride = rays.shape
        nc, nf = opt["num_coarse"], opt["num_fine"]
        cfg = L.RenderCfg(nc, nf, int(bool(opt.get("perturb", True))), int(bool(opt.get("lindisp", False))),
                          int(bool(opt.get("white_background", False))), float(opt.get("noise_std", 0.0)), stride)
        dr = self.dev(np.ascontiguousarray(rays, np.float32))
        dt = self.dev(self._linspace01(nc))
        dud = self.dev(self._linspace01(nf)) if nf > 0 else None
        keep = [self.devopt(rand.get(k)) for k in ("t_rand", "noise_coarse", "u", "noise_fine")]
        rr = L.RenderRand(*[self.p(k) for k in keep])
        names = ("rgb_coarse", "disp_coarse", "acc_coarse", "depth_coarse", "rgb_fine", "disp_fine", "acc_fine", "depth_fine")
        bufs = {k: self.empty((n, 3) if k.startswith("rgb") else (n,)) for k in names}
        ro = L.RenderOut(*[self.ptr(bufs[k]) for k in names])
        wsb = self.lib.render_workspace_bytes(plan_c, plan_f, C.byref(cfg), n, int(training))
        assert wsb >= 0, self.lib.last_error()
        ws = self.empty((wsb // 4 + 1,))
        self.lib.render_fwd(plan_c, plan_f, C.byref(cfg), self.ptr(dr), n, self.ptr(packed_c), self.p(packed_f),
                            self.ptr(dt), self.p(dud), C.byref(rr), seed, ray_offset, C.byref(ro), self.ptr(ws), wsb,
                            int(training), self.stream())
        out = {k: self.host(v) for k, v in bufs.items()}
        for name in want_regions:  # per-sample intermediates the forward left in the workspace
            off, nb = C.c_int64(), C.c_int64()
            self.lib.render_workspace_region(plan_c, plan_f, C.byref(cfg), n, int(training), name.encode(), C.byref(off),
                                             C.byref(nb))
            flat = self.host(ws[off.value // 4:(off.value + nb.value) // 4])
            out[name] = flat.reshape(n, -1, 4) if name.startswith("raw") else flat.reshape(n, -1)
        if nf == 0:
            for k in names[4:]:
                out[k] = None
        if callable(g_rgb):  # g_rgb(outputs) -> (g_coarse, g_fine): the loss between this forward and its backward
            g_rgb = g_rgb(out)
        if g_rgb is not None:
            gc, gf = self.dev(np.ascontiguousarray(g_rgb[0], np.float32)), self.devopt(g_rgb[1])
            gpc = self.empty((self.lib.plan_num_params(plan_c),))
            gpf = self.empty((self.lib.plan_num_params(plan_f),)) if nf > 0 else None
            compacted = bool(self.lib.plan_bwd_compaction(plan_c)) or bool(plan_f and self.lib.plan_bwd_compaction(plan_f))
            if ray_grad_params is None and compacted and int(training) == 1:
                # (one set of backward buffers per net -- the layout the workspace was sized for --, so that each net's kept / total
                # counts survive the other net's backward: nerfhip_render_bwd shares ONE scratch between the two)
                cot = L.RenderCotangents(self.ptr(gc), None, None, self.p(gf), None, None)
                self.lib.render_bwd_parts(plan_c, plan_f, C.byref(cfg), self.ptr(dr), n, self.ptr(packed_c), self.p(packed_f),
                                          C.byref(rr), seed, ray_offset, C.byref(cot), self.ptr(ws), wsb, self.ptr(gpc), self.p(gpf),
                                          L.PART_COARSE | (L.PART_FINE if nf > 0 else 0), self.stream())
            elif ray_grad_params is None:
                self.lib.render_bwd(plan_c, plan_f, C.byref(cfg), self.ptr(dr), n, self.ptr(packed_c), self.p(packed_f),
                                    C.byref(rr), seed, ray_offset, self.ptr(gc), self.p(gf), self.ptr(ws), wsb, self.ptr(gpc),
                                    self.p(gpf), self.stream())
            else:  # ... and d(loss)/d(rays): ray_grad_params = the two flat parameter vectors
                fc, ff = self.dev(ray_grad_params[0]), self.devopt(ray_grad_params[1])
                tb = self.lib.render_bwd_rays_tmp_bytes(plan_c, plan_f, C.byref(cfg), n)
                tmp, g_rays = self.empty((tb // 4 + 1,)), self.empty((n, stride))
                cot = L.RenderCotangents(self.ptr(gc), None, None, self.p(gf), None, None)
                self.lib.render_bwd_rays(plan_c, plan_f, C.byref(cfg), self.ptr(dr), n, self.ptr(packed_c), self.p(packed_f),
                                         C.byref(rr), seed, ray_offset, C.byref(cot), self.ptr(ws), wsb, self.ptr(gpc),
                                         self.p(gpf), L.PART_COARSE | (L.PART_FINE if nf > 0 else 0), self.ptr(fc), self.p(ff),
                                         self.ptr(tmp), tb, self.ptr(g_rays), self.stream())
                out["g_rays"] = self.host(g_rays)
            out["g_params_coarse"] = self.host(gpc)
            out["g_params_fine"] = self.host(gpf) if gpf is not None else None
            # (kept, total) sample points of each net's backward, where its plan runs compacted (nerfhip_plan_set_bwd_compaction)
            for name, plan, samples in (("coarse", plan_c, nc), ("fine", plan_f, nc + nf)):
                if plan is None or samples == 0 or not self.lib.plan_bwd_compaction(plan):
                    continue
                off, nb = C.c_int64(), C.c_int64()
                if not (ray_grad_params is None and int(training) == 1):
                    continue  # (shared backward buffers: the second net's backward has overwritten the first one's counts)
                self.lib.render_workspace_region(plan_c, plan_f, C.byref(cfg), n, 1, ("bwd_scratch_" + name).encode(), C.byref(off),
                                                 C.byref(nb))
                so = self.lib.plan_bwd_stats_offset(plan, n * samples)
                w = self.host(ws[(off.value + so) // 4:(off.value + so) // 4 + 2])
                out["bwd_kept_" + name] = tuple(int(v) for v in np.ascontiguousarray(w).view(np.int32))
        return out

    def mse_loss(self, rgb_c, rgb_f, target, grad_scale=1.0):
        n = rgb_c.shape[0]
        dc, df, dt = self.dev(rgb_c), self.devopt(rgb_f), self.dev(np.ascontiguousarray(target, np.float32))
        gc, gf, lo = self.empty((n, 3)), self.empty((n, 3)), self.empty((3,))
        self.lib.mse_loss_fwd_bwd(self.ptr(dc), self.p(df), self.ptr(dt), target.shape[1], n, float(grad_scale),
                                  self.ptr(gc), self.ptr(gf), self.ptr(lo), self.stream())
        return self.host(lo), self.host(gc), self.host(gf)

    def adam_step(self, p, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
        dp, dg, dm, dv = self.dev(p), self.dev(g), self.dev(m), self.dev(v)
        self.lib.adam_step(self.ptr(dp), self.ptr(dg), self.ptr(dm), self.ptr(dv), p.size, float(lr), float(beta1),
                           float(beta2), float(eps), int(step), float(grad_scale), self.stream())
        return self.host(dp), self.host(dm), self.host(dv)


class EmuBackend(Backend):
    name = "emu"

    def __init__(self):
        self.lib = L.bind(build_emu())
        assert self.lib.is_emulated() == 1

    def dev(self, a):
        return np.array(a, copy=True, order="C")

    def empty(self, shape, dtype=np.float32):
        return np.full(shape, np.nan if dtype == np.float32 else 0, dtype=dtype)

    def ptr(self, d):
        return d.ctypes.data

    def host(self, d):
        return d


class GpuBackend(Backend):
    name = "gpu"

    def __init__(self):
        import torch
        self.torch = torch
        self.lib = L.get_lib()
        assert self.lib.is_emulated() == 0

    def dev(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a)).cuda()

    def empty(self, shape, dtype=np.float32):
        td = {np.float32: self.torch.float32, np.int64: self.torch.int64, np.int32: self.torch.int32,
              np.uint8: self.torch.uint8}[dtype]
        t = self.torch.empty(shape, dtype=td, device="cuda")
        if dtype == np.float32:
            t.fill_(float("nan"))
        return t

    def ptr(self, d):
        return d.data_ptr()

    def host(self, d):
        self.torch.cuda.synchronize()
        return d.cpu().numpy()

    def stream(self):
        return self.torch.cuda.current_stream().cuda_stream
