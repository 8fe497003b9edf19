"""GPU suite (-m gpu), full BASELINE batches against the oracle.

Every ray of a 4096-ray batch of BASELINE configs[1] (lego: 64 coarse + 128 fine, 8x256 nets, noise 0.2) and configs[3]
(fern: NDC rays, 6 xyz frequencies, 64 + 64, 8x128 skip-3 nets, noise 1.0) is rendered through the C ABI with the
oracle's random draws injected and compared with ``oracle.render_rays`` -- outputs AND parameter gradients -- so the
kernels are checked where they actually run: thousands of workgroups, every split-K slice of the weight-gradient
kernel, 24 rounds of the forward grid (nerf/train_utils.py:28-127 is the function restated).

The fine pass sits behind the inverse-CDF sampler, whose conditioning amplifies ulp-level differences of the coarse
weights, so for the fine outputs and gradients the tests assert the documented bounds and RECORD the
measured max / p99.9 (gpurun_out/parity_fullsize_*.json -> profiles/).  ``test_*_teacher_forced_fine_pass`` removes the
sampler from the comparison: the oracle's own z_fine is fed to the unit kernels (MLP forward on oracle-encoded points,
compositing, both backward kernels), which pins the S = 192 fine-net backward at the tight 2e-5 * max|g| bound.
"""
import json
import os
import time

import numpy as np
import pytest
import torch

import nerf_oracle as O
import parity_cases as P

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stats(got, want):
    err = np.abs(np.nan_to_num(np.asarray(got, np.float64)) - np.nan_to_num(np.asarray(want, np.float64))).reshape(-1)
    return dict(max=float(err.max()), p999=float(np.quantile(err, 0.999)), mean=float(err.mean()))


def _grad_stats(got, ref):
    """Per parameter tensor: |got - ref| relative to max|ref| of the tensor; returns the worst tensor's max and p99.9."""
    worst = dict(max=0.0, p999=0.0, tensor="")
    per = {}
    for k, r in ref.items():
        scale = float(np.abs(r).max()) + 1e-30
        e = (np.abs(np.asarray(got[k], np.float64) - np.asarray(r, np.float64)) / scale).reshape(-1)
        per[k] = dict(max=float(e.max()), p999=float(np.quantile(e, 0.999)))
        if per[k]["max"] > worst["max"]:
            worst = dict(max=per[k]["max"], p999=per[k]["p999"], tensor=k)
    return worst, per


def _record(name, payload):
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_fullsize_%s.json" % name), "w") as f:
            json.dump(payload, f, indent=1, sort_keys=True)
    except OSError:
        pass
    print("parity_fullsize_%s: %s" % (name, json.dumps(payload, sort_keys=True)))


def _batch_size(bytes_per_ray, want=4096):
    """The oracle keeps every activation for autograd (~5 MB per ray for 8x256 / 64+128): shrink the batch on a small host."""
    try:
        import psutil
        avail = psutil.virtual_memory().available
    except Exception:
        avail = 64 << 30
    n = want
    while n > 512 and n * bytes_per_ray * 1.6 > avail:
        n //= 2
    return n


class _Case:
    """One oracle run (forward + backward, all rays) shared by the tests of a configuration."""

    def __init__(self, gpu, name, cfg, n, nc, nf, noise, ndc, seed):
        self.gpu, self.name, self.cfg, self.n, self.nc, self.nf = gpu, name, cfg, n, nc, nf
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        g = torch.Generator().manual_seed(seed)
        self.plan_c, par_c, _, self.packed_c = P.mlp_setup(gpu, cfg, seed=seed + 1)
        self.plan_f, par_f, _, self.packed_f = P.mlp_setup(gpu, cfg, seed=seed + 2)
        if ndc:
            H, W, focal = 378, 504, 407.5
            ro = torch.tensor([0.0, 0.0, 0.3]).expand(n, 3) + 0.05 * torch.randn(n, 3, generator=g)
            rd = torch.randn(n, 3, generator=g) * 0.3
            rd[:, 2] = -1.0
            no, nd = O.ndc_rays(H, W, focal, 1.0, ro, rd)
            rays = O.pack_rays(no, nd, 0.0, 1.0, rd)
        else:
            ro = torch.tensor([0.0, 0.0, 4.0]).expand(n, 3) + 0.02 * torch.randn(n, 3, generator=g)
            rd = torch.randn(n, 3, generator=g) * 0.35
            rd[:, 2] = -1.0
            rays = O.pack_rays(ro, rd, 2.0, 6.0, rd)
        self.rays = rays
        self.rand = dict(t_rand=torch.rand(n, nc, generator=g), noise_coarse=torch.randn(n, nc, generator=g),
                         u=torch.rand(n, nf, generator=g), noise_fine=torch.randn(n, nc + nf, generator=g))
        self.opt = dict(num_coarse=nc, num_fine=nf, perturb=True, lindisp=False, white_background=False, noise_std=noise)
        self.tgt = torch.rand(n, 3, generator=g)
        self.par_c = {k: v.requires_grad_(True) for k, v in par_c.items()}
        self.par_f = {k: v.requires_grad_(True) for k, v in par_f.items()}
        t0 = time.perf_counter()
        self.want = O.render_rays(rays, self.par_c, self.par_f, cfg, cfg, self.opt, self.rand, chunksize=131072)
        self.loss, _, _, _ = O.loss_and_psnr(self.want["rgb_coarse"], self.want["rgb_fine"], self.tgt)
        self.loss.backward()
        self.oracle_seconds = time.perf_counter() - t0
        self.ref_gc = {k: v.grad.numpy() for k, v in self.par_c.items()}
        self.ref_gf = {k: v.grad.numpy() for k, v in self.par_f.items()}
        self.rnp = {k: v.numpy() for k, v in self.rand.items()}

    def close(self):
        self.gpu.lib.plan_destroy(self.plan_c)
        self.gpu.lib.plan_destroy(self.plan_f)


@pytest.fixture(scope="module")
def lego(gpu):
    n = _batch_size(6.0e6)
    c = _Case(gpu, "lego_8x256_64+128", P.MLP_GEOMETRIES["northstar8x256"], n, 64, 128, 0.2, False, seed=101)
    yield c
    c.close()


@pytest.fixture(scope="module")
def fern(gpu):
    n = _batch_size(2.5e6)
    c = _Case(gpu, "fern_8x128_ndc_64+64", P.MLP_GEOMETRIES["fern8x128_skip3_L6"], n, 64, 64, 1.0, True, seed=202)
    yield c
    c.close()


@pytest.fixture(scope="module")
def lego_default_nets(gpu):
    """The nets the reference's scripts really build (FlexibleNeRFModel defaults: 4 x 128, skip 4 -- SURVEY 0.2) on the
    lego batch geometry."""
    n = _batch_size(3.0e6)
    c = _Case(gpu, "lego_4x128_64+128", P.MLP_GEOMETRIES["default4x128"], n, 64, 128, 0.2, False, seed=303)
    yield c
    c.close()


@pytest.fixture(scope="module")
def lego_padded_nets(gpu):
    """A hidden size between the kernel widths (5 x 99, skip 2: rides zero-padded on the 128-wide kernels), 2048 rays."""
    c = _Case(gpu, "lego_5x99_64+128", P.MLP_GEOMETRIES["odd5x99_skip2"], 2048, 64, 128, 0.2, False, seed=404)
    yield c
    c.close()


def _end_to_end(c, coarse_grad_tol, fine_grad_tol, rgb_fine_tol=(1e-4, 1e-4)):
    gpu = c.gpu
    out = gpu.render(c.plan_c, c.plan_f, c.packed_c, c.packed_f, c.rays.numpy(), c.opt, c.rnp, training=True)
    l3, gc, gf = gpu.mse_loss(out["rgb_coarse"], out["rgb_fine"], c.tgt.numpy())
    out2 = gpu.render(c.plan_c, c.plan_f, c.packed_c, c.packed_f, c.rays.numpy(), c.opt, c.rnp, training=True, g_rgb=(gc, gf))
    w = {k: v.detach().numpy() for k, v in c.want.items() if v is not None}
    rec = dict(rays=c.n, samples="%d+%d" % (c.nc, c.nf), oracle_seconds=round(c.oracle_seconds, 1),
               oracle_threads=torch.get_num_threads(), outputs={}, loss=dict(gpu=float(l3[2]), oracle=float(c.loss.detach())))
    for k in ("rgb_coarse", "acc_coarse", "depth_coarse", "rgb_fine", "acc_fine", "depth_fine"):
        rec["outputs"][k] = _stats(out[k], w[k])
    for k in ("disp_coarse", "disp_fine"):  # NaN where acc == 0 (volume_rendering_utils.py:48): same pixels on both sides
        assert np.array_equal(np.isnan(out[k]), np.isnan(w[k])), k
        rel = np.abs(np.nan_to_num(out[k]) - np.nan_to_num(w[k])) / (1.0 + np.abs(np.nan_to_num(w[k])))
        rec["outputs"][k + "_rel"] = dict(max=float(rel.max()), p999=float(np.quantile(rel, 0.999)))
    gcw, gcp = _grad_stats(gpu.unflatten(c.plan_c, out2["g_params_coarse"]), c.ref_gc)
    gfw, gfp = _grad_stats(gpu.unflatten(c.plan_f, out2["g_params_fine"]), c.ref_gf)
    rec["grad_coarse_worst_rel"] = gcw
    rec["grad_fine_worst_rel"] = gfw
    rec["grad_fine_per_tensor"] = gfp
    _record(c.name, rec)
    # coarse pass: fp32 round-off only
    for k in ("rgb_coarse", "acc_coarse", "depth_coarse"):
        assert rec["outputs"][k]["max"] <= 1e-5, (k, rec["outputs"][k])
    # the north-star bar on colour; acc / depth of the fine pass carry the sampler's conditioning
    assert rec["outputs"]["rgb_fine"]["max"] <= rgb_fine_tol[0] and rec["outputs"]["rgb_fine"]["p999"] <= rgb_fine_tol[1], \
        rec["outputs"]["rgb_fine"]
    assert rec["outputs"]["acc_fine"]["max"] <= 5e-4 and rec["outputs"]["depth_fine"]["max"] <= 2e-3, rec["outputs"]
    assert abs(float(l3[2]) - float(c.loss)) < 1e-5
    assert gcw["max"] <= coarse_grad_tol[0] and gcw["p999"] <= coarse_grad_tol[1], gcw
    assert gfw["max"] <= fine_grad_tol[0] and gfw["p999"] <= fine_grad_tol[1], gfw
    return rec


def test_lego_full_batch_every_ray_vs_oracle(lego):
    """BASELINE configs[1]: outputs of all rays and all 2 x 595,844 gradient entries against the oracle.
    Measured on MI355X (profiles/r02_parity_fullsize.json): rgb_fine max 7.9e-5 / p99.9 4.7e-5; coarse-net gradients
    max 1.4e-4 / p99.9 3.6e-5 of max|g| (two fp32 sums of 262,144 terms in different orders); fine-net gradients max
    6.5e-4 / p99.9 3.6e-4 (behind the sampler)."""
    _end_to_end(lego, coarse_grad_tol=(2e-4, 5e-5), fine_grad_tol=(5e-3, 1e-3))


def test_lego_default_4x128_nets_full_batch_vs_oracle(lego_default_nets):
    _end_to_end(lego_default_nets, coarse_grad_tol=(2e-4, 5e-5), fine_grad_tol=(5e-3, 1e-3))


def test_lego_padded_hidden_size_batch_vs_oracle(lego_padded_nets):
    """Not a BASELINE configuration: the coarse pass (no sampler in front of it) is held to the same bounds as above;
    behind the sampler ONE ray of 2048 exceeds the 1e-4 colour bar of the BASELINE configurations (measured max 1.6e-4,
    p99.9 7.6e-5 -- a fine sample that lands in a neighbouring bin), so this case asserts p99.9 <= 1e-4 and max <= 3e-4;
    the teacher-forced fine pass below pins the kernels themselves."""
    _end_to_end(lego_padded_nets, coarse_grad_tol=(2e-4, 5e-5), fine_grad_tol=(5e-3, 1e-3), rgb_fine_tol=(3e-4, 1e-4))


def test_lego_padded_hidden_size_teacher_forced_fine_pass(lego_padded_nets):
    _teacher_forced(lego_padded_nets)


def test_fern_full_batch_every_ray_vs_oracle(fern):
    """BASELINE configs[3] (NDC, Dx = 39, 64 + 64, noise 1.0): with sigma noise of std 1.0 the per-sample cotangents of
    the early layers nearly cancel, hence the wider coarse-gradient bound (see case_render_vs_oracle)."""
    _end_to_end(fern, coarse_grad_tol=(1e-3, 3e-4), fine_grad_tol=(3e-2, 1e-2))


def _fine_pass_units(c, sel, z, tgt):
    """The fine pass of rays `sel` with given depths through the unit entry points of the C ABI: MLP forward on
    host-encoded points (writes the stash) -> compositing -> compositing backward -> MLP backward."""
    gpu, cfg = c.gpu, c.cfg
    rays = c.rays[sel]
    n, s = z.shape
    ro, rd = rays[..., :3], rays[..., 3:6]
    pts = (ro[..., None, :] + rd[..., None, :] * z[..., :, None]).reshape(-1, 3)
    emb = O.positional_encoding(pts, cfg["num_encoding_fn_xyz"], True, True)
    dirs = rays[..., None, -3:].expand(n, s, 3).reshape(-1, 3)
    x = torch.cat((emb, O.positional_encoding(dirs, cfg["num_encoding_fn_dir"], True, True)), dim=-1).numpy()
    raw, stash = gpu.mlp_fwd(c.plan_f, c.packed_f, x, want_stash=True)
    noise = c.rnp["noise_fine"][sel]
    rgb, disp, acc, w, dep = gpu.volume_render_fwd(raw.reshape(n, s, 4), z.numpy(), rd.numpy(), c.opt["noise_std"], noise)
    g_rgb = ((2.0 / (3.0 * n)) * (rgb - tgt.numpy())).astype(np.float32)  # d mse_loss / d rgb  (train_nerf.py:250-258)
    g_raw = gpu.volume_render_bwd(raw.reshape(n, s, 4), z.numpy(), rd.numpy(), g_rgb=g_rgb, noise_std=c.opt["noise_std"],
                                  noise=noise)
    gflat = gpu.mlp_bwd(c.plan_f, c.packed_f, g_raw.reshape(-1, 4), stash)
    return raw, rgb, acc, gpu.unflatten(c.plan_f, gflat)


def _oracle_fine_grads(c, sel, z, tgt, dtype):
    """The same pass on the oracle in `dtype` (fp64 = the yardstick both fp32 implementations are measured against)."""
    par = {k: v.detach().to(dtype).requires_grad_(True) for k, v in c.par_f.items()}
    rays = c.rays[sel].to(dtype)
    ro, rd = rays[..., :3], rays[..., 3:6]
    pts = ro[..., None, :] + rd[..., None, :] * z.to(dtype)[..., :, None]
    raw = O.run_network(par, pts, rays, c.cfg)
    rgb = O.volume_render(raw, z.to(dtype), rd, c.opt["noise_std"], c.rand["noise_fine"][sel].to(dtype))[0]
    torch.nn.functional.mse_loss(rgb, tgt.to(dtype)).backward()
    return {k: v.grad.numpy() for k, v in par.items()}


def _teacher_forced(c):
    """The fine pass with the ORACLE's depths (no sampler between the two sides).  Full batch against the oracle's fp32
    gradients, then a 256-ray slice against an fp64 run of the oracle: the kernels must sit at the fp32 floor, i.e. no
    further from fp64 than torch's own fp32 path is.  (What remains between two fp32 implementations is not only the
    order of the 786,432-term sums: a pre-activation within an ulp of zero takes the other branch of a ReLU, and with
    ~1e9 pre-activations per batch a handful do.  On the small slice one such sample is visible -- measured: both fp32
    paths are 2.6e-3 of max|g| away from fp64 in the SAME entry -- hence quantiles, not maxima, on the slice.)"""
    n = c.n
    z = c.want["z_fine"].detach()
    raw, rgb, acc, grads = _fine_pass_units(c, slice(0, n), z, c.tgt)
    rec = dict(rays=n, samples_per_ray=c.nc + c.nf, raw=_stats(raw, c.want["raw_fine"].detach().numpy().reshape(-1, 4)),
               rgb_fine=_stats(rgb, c.want["rgb_fine"].detach().numpy()),
               acc_fine=_stats(acc, c.want["acc_fine"].detach().numpy()))
    worst, per = _grad_stats(grads, c.ref_gf)
    rec["grad_fine_worst_rel"] = worst
    rec["grad_fine_per_tensor"] = per
    m = 256
    sel = slice(0, m)
    _, _, _, g_hip = _fine_pass_units(c, sel, z[sel], c.tgt[sel])
    g32 = _oracle_fine_grads(c, sel, z[sel], c.tgt[sel], torch.float32)
    g64 = _oracle_fine_grads(c, sel, z[sel], c.tgt[sel], torch.float64)
    rec["slice_rays"] = m
    rec["slice_hip_vs_fp64"] = _grad_stats(g_hip, g64)[0]
    rec["slice_torch_fp32_vs_fp64"] = _grad_stats(g32, g64)[0]
    rec["slice_hip_vs_torch_fp32"] = _grad_stats(g_hip, g32)[0]
    _record(c.name + "_teacher_forced", rec)
    assert rec["raw"]["max"] <= 1e-6, rec["raw"]
    assert rec["rgb_fine"]["max"] <= 2e-6 and rec["acc_fine"]["max"] <= 2e-6, rec
    # a gradient entry is a sum over 786,432 (lego) samples: the two fp32 summation orders differ by ~sqrt(N) eps
    assert worst["max"] <= 1e-4 and worst["p999"] <= 5e-5, worst
    assert rec["slice_hip_vs_fp64"]["p999"] <= 1.5 * rec["slice_torch_fp32_vs_fp64"]["p999"] + 1e-6, rec
    assert rec["slice_hip_vs_fp64"]["max"] <= 1.5 * rec["slice_torch_fp32_vs_fp64"]["max"] + 1e-4, rec


def test_lego_teacher_forced_fine_pass(lego):
    _teacher_forced(lego)


def test_fern_teacher_forced_fine_pass(fern):
    _teacher_forced(fern)


# ---- BASELINE configs[0]: tiny_nerf.py (100x100, 32 samples, coarse only, 6 frequencies, no view directions) ---------------
def _tiny_nerf_iteration(be, H, W, focal, pose, w1, b1, w2, b2, w3, b3, noise, target):
    """run_one_iter_of_tinynerf (tiny_nerf.py:111-159) + the loss of its training loop (:293-299), written once over the
    four helpers tiny_nerf.py imports (:9): `be` provides get_ray_bundle / positional_encoding / cumprod_exclusive /
    get_minibatches -- this package's HIP versions or the oracle's."""
    ro, rd = be.get_ray_bundle(H, W, focal, pose)
    near, far, ns = 2.0, 6.0, 32
    depth = torch.linspace(near, far, ns).to(ro)                                  # compute_query_points_from_rays :44-58
    depth = depth + noise * (far - near) / ns
    pts = ro[..., None, :] + rd[..., None, :] * depth[..., :, None]
    flat = pts.reshape((-1, 3))
    enc = be.positional_encoding(flat, 6)
    preds = []
    for batch in be.get_minibatches(enc, chunksize=16384):                       # :139-144
        h = torch.relu(torch.nn.functional.linear(batch, w1, b1))                # VeryTinyNerfModel :160-176
        h = torch.relu(torch.nn.functional.linear(h, w2, b2))
        preds.append(torch.nn.functional.linear(h, w3, b3))
    rf = torch.cat(preds, dim=0).reshape(list(pts.shape[:-1]) + [4])
    sigma_a = torch.relu(rf[..., 3])                                             # render_volume_density :83-108
    rgb = torch.sigmoid(rf[..., :3])
    one_e_10 = torch.tensor([1e10]).to(ro)
    dists = torch.cat((depth[..., 1:] - depth[..., :-1], one_e_10.expand(depth[..., :1].shape)), dim=-1)
    alpha = 1.0 - torch.exp(-sigma_a * dists)
    weights = alpha * be.cumprod_exclusive(1.0 - alpha + 1e-10)
    rgb_map = (weights[..., None] * rgb).sum(dim=-2)
    return rgb_map, torch.nn.functional.mse_loss(rgb_map, target)


def test_tiny_nerf_geometry_through_the_helpers_vs_oracle():
    """BASELINE configs[0]: one tiny_nerf.py training iteration (forward, loss, backward into a user torch model) built
    on this package's helpers on the GPU equals the same composition on the oracle's helpers on the CPU."""
    import types

    import nerf_pytorch_amd as N
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    dev = torch.device("cuda", 0)
    H = W = 100
    focal = 138.88887889922103  # tiny_nerf_data.npz's focal length for 100x100 images
    g = torch.Generator().manual_seed(9458)
    pose = torch.eye(4)
    pose[:3, :3] = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
    pose[:3, 3] = torch.tensor([0.3, -0.2, 4.0])
    shapes = [(128, 39), (128,), (128, 128), (128,), (4, 128), (4,)]
    params = [(torch.rand(s, generator=g) * 2 - 1) * (1.0 / np.sqrt(s[-1] if len(s) > 1 else 128)) for s in shapes]
    noise = torch.rand(H, W, 32, generator=g)
    target = torch.rand(H, W, 3, generator=g)
    cpu_be = types.SimpleNamespace(get_ray_bundle=O.get_ray_bundle, cumprod_exclusive=O.cumprod_exclusive,
                                   positional_encoding=lambda x, n: O.positional_encoding(x, n, True, True),
                                   get_minibatches=N.get_minibatches)
    gpu_be = types.SimpleNamespace(get_ray_bundle=N.get_ray_bundle, cumprod_exclusive=N.cumprod_exclusive,
                                   positional_encoding=lambda x, n: N.positional_encoding(x, num_encoding_functions=n),
                                   get_minibatches=N.get_minibatches)
    pc = [p.clone().requires_grad_(True) for p in params]
    pg = [p.clone().to(dev).requires_grad_(True) for p in params]
    rgb_c, loss_c = _tiny_nerf_iteration(cpu_be, H, W, focal, pose, *pc, noise, target)
    rgb_g, loss_g = _tiny_nerf_iteration(gpu_be, H, W, focal, pose.to(dev), *pg, noise.to(dev), target.to(dev))
    loss_c.backward()
    loss_g.backward()
    # the user's network runs on torch's own GEMMs on both sides (rocBLAS vs MKL): fp32 round-off of a 3-layer MLP
    P.close(rgb_g.detach().cpu().numpy(), rgb_c.detach().numpy(), 2e-5, what="tiny_nerf rgb")
    assert abs(float(loss_g) - float(loss_c)) < 1e-6
    for a, b in zip(pg, pc):
        ref = b.grad.numpy()
        scale = float(np.abs(ref).max()) + 1e-12
        P.close(a.grad.cpu().numpy(), ref, 2e-4 * scale + 1e-9, 1e-3, what="tiny_nerf grad")
