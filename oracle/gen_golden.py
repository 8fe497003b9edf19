"""gen_golden.py -- TEST INFRASTRUCTURE, build-container only.

Runs the REAL reference (/root/reference, imported via ref_import.py) on seeded inputs and writes the outputs as small
fixtures under tests/golden/.  The fixtures travel to the GPU box; /root/reference does not.  Re-run with
    python oracle/gen_golden.py
whenever a case is added.  Random draws the reference makes (torch.rand / torch.randn) are injected and recorded.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import nerf_oracle as O  # noqa: E402
import ref_import as R  # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")


def npz(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(GOLD, name)
    np.savez_compressed(path, **out)
    print("wrote %-28s %8.1f KB" % (name, os.path.getsize(path) / 1024))


def pose_like(g):
    """A rigid camera-to-world matrix from a seeded generator (rotation via QR)."""
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g))
    if torch.det(q) < 0:
        q[:, 0] = -q[:, 0]
    c2w = torch.eye(4)
    c2w[:3, :3] = q
    c2w[:3, 3] = torch.randn(3, generator=g) * 2
    return c2w


def helpers(nerf):
    g = torch.Generator().manual_seed(7)
    d = {}
    # get_ray_bundle
    c2w = pose_like(g)
    ro, rd = nerf.get_ray_bundle(5, 7, 3.3, c2w)
    d.update(rb_c2w=c2w, rb_ro=ro.contiguous(), rb_rd=rd)
    c2w2 = pose_like(g)
    ro2, rd2 = nerf.get_ray_bundle(20, 16, 555.5555 / 20, c2w2[:3, :4])
    d.update(rb2_c2w=c2w2[:3, :4], rb2_ro=ro2.contiguous(), rb2_rd=rd2)
    # ndc_rays
    o = torch.randn(33, 3, generator=g)
    o[:, 2] = o[:, 2].abs() + 1.5
    dd = torch.randn(33, 3, generator=g)
    dd[:, 2] = -dd[:, 2].abs() - 0.3
    no, nd = nerf.ndc_rays(378, 504, 407.5, 1.0, o, dd)
    d.update(ndc_o=o, ndc_d=dd, ndc_out_o=no, ndc_out_d=nd)
    # positional encoding
    x = torch.randn(41, 3, generator=g) * 3.0
    d["pe_x"] = x
    d["pe_L10"] = nerf.positional_encoding(x, 10, True, True)
    d["pe_L4"] = nerf.positional_encoding(x, 4, True, True)
    d["pe_L6_noinput"] = nerf.positional_encoding(x, 6, False, True)
    d["pe_L3_linear"] = nerf.positional_encoding(x, 3, True, False)
    d["pe_L0"] = nerf.positional_encoding(x, 0, True, True)
    # cumprod_exclusive
    cp = torch.rand(9, 70, generator=g)
    d.update(cp_x=cp, cp_y=nerf.cumprod_exclusive(cp))
    # volume rendering: with noise + white background, and plain
    raw = torch.randn(6, 40, 4, generator=g) * 2
    z = torch.sort(torch.rand(6, 40, generator=g) * 4 + 2, dim=-1)[0]
    rdv = torch.randn(6, 3, generator=g)
    nz = torch.randn(6, 40, generator=g)
    with R.injected_randoms([nz]):
        r1 = nerf.volume_render_radiance_field(raw, z, rdv, radiance_field_noise_std=0.7, white_background=True)
    r2 = nerf.volume_render_radiance_field(raw, z, rdv, radiance_field_noise_std=0.0, white_background=False)
    d.update(vr_raw=raw, vr_z=z, vr_rd=rdv, vr_noise=nz)
    for i, n in enumerate(("rgb", "disp", "acc", "weights", "depth")):
        d["vr1_" + n] = r1[i]
        d["vr2_" + n] = r2[i]
    # a ray that hits nothing: acc == 0 -> NaN disparity
    raw0 = raw.clone()
    raw0[0, :, 3] = -5.0
    r3 = nerf.volume_render_radiance_field(raw0, z, rdv)
    d.update(vr3_raw=raw0, vr3_disp=r3[1], vr3_acc=r3[2], vr3_rgb=r3[0])
    # sample_pdf_2: random u and deterministic
    bins = torch.sort(torch.rand(11, 63, generator=g) * 4 + 2, dim=-1)[0]
    w = torch.rand(11, 62, generator=g) ** 4
    w[3] = 0.0          # all-flat pdf
    w[4, 10:50] = 0.0   # plateaus
    u = torch.rand(11, 128, generator=g)
    with R.injected_randoms([u]):
        s_rand = nerf.sample_pdf_2(bins, w, 128, det=False)
    s_det = nerf.sample_pdf_2(bins, w, 128, det=True)
    d.update(sp_bins=bins, sp_w=w, sp_u=u, sp_rand=s_rand, sp_det=s_det)
    npz("helpers.npz", **d)


def make_opts(nerf, num_coarse, num_fine, perturb, lindisp, white, noise, no_ndc=True, near=2.0, far=6.0, viewdirs=True):
    blk = dict(num_random_rays=1024, chunksize=1 << 17, perturb=perturb, num_coarse=num_coarse, num_fine=num_fine,
               white_background=white, radiance_field_noise_std=noise, lindisp=lindisp)
    return nerf.CfgNode({"nerf": {"use_viewdirs": viewdirs, "train": dict(blk), "validation": dict(blk)},
                         "dataset": {"no_ndc": no_ndc, "near": near, "far": far}})


def e2e_case(nerf, name, cfg_c, cfg_f, n_rays, nc, nf, perturb, lindisp, white, noise, seed, ndc=False):
    g = torch.Generator().manual_seed(seed)
    mc = R.make_reference_model(nerf, cfg_c)
    mf = R.make_reference_model(nerf, cfg_f)
    pc = O.init_params(cfg_c, seed=seed * 2 + 1)
    pf = O.init_params(cfg_f, seed=seed * 2 + 2)
    mc.load_state_dict(pc)
    mf.load_state_dict(pf)
    view = cfg_c.get("use_viewdirs", True)
    ex = nerf.get_embedding_function(cfg_c["num_encoding_fn_xyz"], cfg_c.get("include_input_xyz", True), True)
    ed = nerf.get_embedding_function(cfg_c["num_encoding_fn_dir"], cfg_c.get("include_input_dir", True), True) if view else None
    H, W, focal = 40, 50, 45.0
    c2w = pose_like(g)
    c2w[:3, 3] = torch.tensor([0.3, -0.2, 4.0])
    if ndc:
        c2w[:3, :3] = torch.eye(3) + 0.05 * torch.randn(3, 3, generator=g)
    ro_img, rd_img = nerf.get_ray_bundle(H, W, focal, c2w)
    pix = torch.randperm(H * W, generator=g)[:n_rays]
    ro = ro_img.reshape(-1, 3)[pix].contiguous()
    rd = rd_img.reshape(-1, 3)[pix].contiguous()
    tgt = torch.rand(n_rays, 3, generator=g)
    near, far = (0.0, 1.0) if ndc else (2.0, 6.0)
    opts = make_opts(nerf, nc, nf, perturb, lindisp, white, noise, no_ndc=not ndc, near=near, far=far, viewdirs=view)
    draws, rec = [], {}
    if perturb:
        rec["t_rand"] = torch.rand(n_rays, nc, generator=g)
        draws.append(rec["t_rand"])
    if noise > 0:
        rec["noise_coarse"] = torch.randn(n_rays, nc, generator=g)
        draws.append(rec["noise_coarse"])
    if perturb:
        rec["u"] = torch.rand(n_rays, nf, generator=g)
        draws.append(rec["u"])
    if noise > 0:
        rec["noise_fine"] = torch.randn(n_rays, nc + nf, generator=g)
        draws.append(rec["noise_fine"])
    with R.injected_randoms(draws):
        out = nerf.run_one_iter_of_nerf(H, W, focal, mc, mf, ro, rd, opts, mode="train", encode_position_fn=ex,
                                        encode_direction_fn=ed)
    loss = torch.nn.functional.mse_loss(out[0], tgt) + torch.nn.functional.mse_loss(out[3], tgt)
    loss.backward()
    d = dict(H=H, W=W, focal=focal, c2w=c2w, pix=pix, ro=ro, rd=rd, target=tgt, near=near, far=far, loss=loss)
    d.update(rec)
    for i, n in enumerate(("rgb_coarse", "disp_coarse", "acc_coarse", "rgb_fine", "disp_fine", "acc_fine")):
        d[n] = out[i]
    for tag, m in (("gc_", mc), ("gf_", mf)):
        for k, p in m.named_parameters():
            d[tag + k] = p.grad
    meta = dict(cfg_c=cfg_c, cfg_f=cfg_f, n_rays=n_rays, nc=nc, nf=nf, perturb=perturb, lindisp=lindisp, white=white,
                noise=noise, seed=seed, ndc=ndc)
    d["meta"] = np.array(repr(meta))
    npz(name, **d)


def e2e_sampled(nerf, name, cfg, n_rays, nc, nf, seed):
    """The HEADLINE geometry (8x256, 64 + 128 samples) against the real reference.  Full gradients would be 4.8 MB, so the
    fixture keeps the outputs, every parameter tensor's gradient sum and absolute sum, and 96 sampled gradient entries
    per tensor (positions recorded in the file).  Weights are oracle init_params(seed), as in e2e_case."""
    g = torch.Generator().manual_seed(seed)
    mc, mf = R.make_reference_model(nerf, cfg), R.make_reference_model(nerf, cfg)
    mc.load_state_dict(O.init_params(cfg, seed=seed * 2 + 1))
    mf.load_state_dict(O.init_params(cfg, seed=seed * 2 + 2))
    ex = nerf.get_embedding_function(cfg["num_encoding_fn_xyz"], True, True)
    ed = nerf.get_embedding_function(cfg["num_encoding_fn_dir"], True, True)
    H, W, focal = 40, 50, 45.0
    c2w = pose_like(g)
    c2w[:3, 3] = torch.tensor([0.3, -0.2, 4.0])
    ro_img, rd_img = nerf.get_ray_bundle(H, W, focal, c2w)
    pix = torch.randperm(H * W, generator=g)[:n_rays]
    ro, rd = ro_img.reshape(-1, 3)[pix].contiguous(), rd_img.reshape(-1, 3)[pix].contiguous()
    tgt = torch.rand(n_rays, 3, generator=g)
    opts = make_opts(nerf, nc, nf, True, False, False, 0.2)
    rec = dict(t_rand=torch.rand(n_rays, nc, generator=g), noise_coarse=torch.randn(n_rays, nc, generator=g),
               u=torch.rand(n_rays, nf, generator=g), noise_fine=torch.randn(n_rays, nc + nf, generator=g))
    with R.injected_randoms([rec["t_rand"], rec["noise_coarse"], rec["u"], rec["noise_fine"]]):
        out = nerf.run_one_iter_of_nerf(H, W, focal, mc, mf, ro, rd, opts, mode="train", encode_position_fn=ex,
                                        encode_direction_fn=ed)
    loss = torch.nn.functional.mse_loss(out[0], tgt) + torch.nn.functional.mse_loss(out[3], tgt)
    loss.backward()
    d = dict(H=H, W=W, focal=focal, c2w=c2w, pix=pix, ro=ro, rd=rd, target=tgt, near=2.0, far=6.0, loss=loss)
    d.update(rec)
    for i, n in enumerate(("rgb_coarse", "disp_coarse", "acc_coarse", "rgb_fine", "disp_fine", "acc_fine")):
        d[n] = out[i]
    pick = np.random.RandomState(seed)
    for tag, m in (("gc_", mc), ("gf_", mf)):
        for k, prm in m.named_parameters():
            gr = prm.grad.reshape(-1)
            idx = pick.randint(0, gr.numel(), size=96).astype(np.int64)
            d["i" + tag + k] = idx
            d["v" + tag + k] = gr[torch.from_numpy(idx)]
            d["s" + tag + k] = torch.stack([gr.sum(), gr.abs().sum(), gr.abs().max()])
    meta = dict(cfg_c=cfg, cfg_f=cfg, n_rays=n_rays, nc=nc, nf=nf, perturb=True, lindisp=False, white=False, noise=0.2,
                seed=seed, ndc=False)
    d["meta"] = np.array(repr(meta))
    npz(name, **d)


def mlp_case(nerf):
    """FlexibleNeRFModel.forward on random encoded rows, incl. the skip geometries of SURVEY 0.3."""
    d = {}
    g = torch.Generator().manual_seed(11)
    for tag, (L, Wd, sk, lx, ld, view) in {"a": (8, 128, 4, 10, 4, True), "b": (8, 128, 3, 6, 4, True),
                                           "c": (6, 128, 2, 10, 4, True), "d": (4, 128, 4, 10, 4, False),
                                           "e": (2, 128, 4, 4, 2, True)}.items():
        cfg = dict(num_layers=L, hidden_size=Wd, skip_connect_every=sk, num_encoding_fn_xyz=lx, num_encoding_fn_dir=ld,
                   use_viewdirs=view)
        m = R.make_reference_model(nerf, cfg)
        m.load_state_dict(O.init_params(cfg, seed=100 + ord(tag)))
        dx, dd = O.model_dims(cfg)
        x = torch.randn(37, dx + dd, generator=g)
        d["x_" + tag] = x
        d["y_" + tag] = m(x)
    npz("mlp_forward.npz", **d)


def pretrained_lego(nerf):
    ck_path = os.path.join(R.REFERENCE_ROOT, "pretrained", "lego-lowres", "checkpoint199999.ckpt")
    ck = torch.load(ck_path, map_location="cpu", weights_only=False)
    cfg = dict(num_layers=4, hidden_size=128, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)
    mc = R.make_reference_model(nerf, cfg)
    mf = R.make_reference_model(nerf, cfg)
    mc.load_state_dict(ck["model_coarse_state_dict"])
    mf.load_state_dict(ck["model_fine_state_dict"])
    w = {"c_" + k: v for k, v in ck["model_coarse_state_dict"].items()}
    w.update({"f_" + k: v for k, v in ck["model_fine_state_dict"].items()})
    npz("lego_lowres_weights.npz", **w)
    # pose_spherical(30, -30, 4) as load_blender.py:32-37 builds it, computed by the reference loader module's helper
    from nerf.load_blender import pose_spherical
    pose = torch.as_tensor(np.asarray(pose_spherical(30.0, -30.0, 4.0)), dtype=torch.float32)
    H = W = 100
    focal = 0.5 * W / np.tan(0.5 * 0.6911112070083618)
    opts = make_opts(nerf, 64, 64, False, False, True, 0.0)
    ex = nerf.get_embedding_function(10, True, True)
    ed = nerf.get_embedding_function(4, True, True)
    ro_img, rd_img = nerf.get_ray_bundle(H, W, focal, pose[:3, :4])
    with torch.no_grad():
        full = nerf.run_one_iter_of_nerf(H, W, focal, mc, mf, ro_img, rd_img, opts, mode="validation",
                                         encode_position_fn=ex, encode_direction_fn=ed)
    stats = dict(rgb_fine_mean=float(full[3].mean()), acc_fine_mean=float(full[5].mean()),
                 acc_gt_half=float((full[5] > 0.5).float().mean()))
    print("pretrained lego 100x100:", stats)
    rows = torch.arange(10, 90, 4)  # a 20 x 100 strip subset keeps the fixture small
    d = dict(pose=pose, H=H, W=W, focal=focal, rows=rows)
    for i, n in enumerate(("rgb_coarse", "disp_coarse", "acc_coarse", "rgb_fine", "disp_fine", "acc_fine")):
        d[n] = full[i][rows]
    d.update({k: np.float64(v) for k, v in stats.items()})
    npz("lego_lowres_render.npz", **d)


def reference_script_function(script, name, namespace):
    """Execute ONE function definition of a reference script (the scripts themselves cannot be imported here:
    torchvision / tensorboard / imageio are absent) and return it."""
    import ast
    path = os.path.join(R.REFERENCE_ROOT, script)
    tree = ast.parse(open(path).read(), path)
    node = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name][0]
    exec(compile(ast.Module([node], []), path, "exec"), namespace)
    return namespace[name]


def dataio(nerf):
    """Rows either side of the path: the training loop's ray selection (train_nerf.py:210-227; those statements sit in
    the script's main() and are restated here literally around the reference's own get_ray_bundle / meshgrid_xy) and
    eval_nerf.py's 8-bit casts (executed from the reference source)."""
    g = torch.Generator().manual_seed(21)
    out = {}
    for tag, (H, W, C, n) in {"a": (5, 7, 4, 12), "b": (20, 16, 3, 64)}.items():
        focal = 0.9 * W
        pose = pose_like(g)
        img = torch.rand(H, W, C, generator=g)
        select_inds = np.random.RandomState(5).choice(H * W, size=(n), replace=False)
        ray_origins, ray_directions = nerf.get_ray_bundle(H, W, focal, pose[:3, :4])
        coords = torch.stack(nerf.meshgrid_xy(torch.arange(H), torch.arange(W)), dim=-1)
        coords = coords.reshape((-1, 2))
        sel = coords[select_inds]
        out.update({"sel_%s_hwfc" % tag: np.array([H, W, focal, C], np.float64), "sel_%s_pose" % tag: pose,
                    "sel_%s_img" % tag: img, "sel_%s_inds" % tag: select_inds.astype(np.int64),
                    "sel_%s_ro" % tag: ray_origins[sel[:, 0], sel[:, 1], :],
                    "sel_%s_rd" % tag: ray_directions[sel[:, 0], sel[:, 1], :],
                    "sel_%s_target" % tag: img[sel[:, 0], sel[:, 1], :]})
    ns = {"np": np, "torch": torch}
    disp_fn = reference_script_function("eval_nerf.py", "cast_to_disparity_image", ns)
    d0 = torch.rand(9, 11, generator=g) * 3 + 0.1
    d1 = d0.clone()
    d1[2, 3] = float("nan")
    d2 = torch.full((4, 4), 0.7)
    d3 = d0.clone()
    d3[0, 0] = float("inf")
    with np.errstate(invalid="ignore"):
        for i, d in enumerate((d0, d1, d2, d3)):
            out["disp%d_in" % i] = d
            out["disp%d_out" % i] = disp_fn(d)
    # cast_to_image needs torchvision (absent): ToPILImage is stubbed with torchvision's published conversion for float
    # tensors (functional.to_pil_image: pic.mul(255).byte(), CHW -> HWC) -- "parity unpinned" for that dependency.
    import types
    from PIL import Image
    tv = types.SimpleNamespace(transforms=types.SimpleNamespace(ToPILImage=lambda: (
        lambda pic: Image.fromarray(np.transpose(pic.mul(255).byte().numpy(), (1, 2, 0)), mode="RGB"))))
    img_fn = reference_script_function("eval_nerf.py", "cast_to_image", dict(ns, torchvision=tv))
    rgb = torch.rand(6, 5, 3, generator=g)
    rgb[0, 0] = torch.tensor([0.0, 1.0, 1.0000001])
    rgb[0, 1] = torch.tensor([254.999 / 255, 0.5, 1 / 255])
    out["img_in"] = rgb
    out["img_out"] = img_fn(rgb, "blender")
    npz("dataio.npz", **out)


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    nerf = R.import_reference()
    if sys.argv[1:] == ["dataio"]:
        return dataio(nerf)
    north = dict(num_layers=8, hidden_size=256, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)
    if sys.argv[1:] == ["northstar"]:
        return e2e_sampled(nerf, "e2e_northstar.npz", north, 5, 64, 128, seed=6)
    helpers(nerf)
    mlp_case(nerf)
    base = dict(num_layers=4, hidden_size=128, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)
    deep = dict(num_layers=8, hidden_size=128, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)
    fern = dict(num_layers=8, hidden_size=128, skip_connect_every=3, num_encoding_fn_xyz=6, num_encoding_fn_dir=4)
    novw = dict(num_layers=4, hidden_size=128, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4,
                use_viewdirs=False)
    e2e_case(nerf, "e2e_a.npz", base, base, 8, 16, 16, True, False, False, 0.2, seed=1)
    e2e_case(nerf, "e2e_b.npz", deep, deep, 3, 64, 128, True, False, True, 1.0, seed=2)
    e2e_case(nerf, "e2e_c.npz", fern, fern, 4, 32, 32, False, False, False, 0.0, seed=3, ndc=True)
    e2e_case(nerf, "e2e_d.npz", novw, novw, 5, 8, 8, True, True, True, 1.0, seed=4)
    pretrained_lego(nerf)
    dataio(nerf)
    e2e_sampled(nerf, "e2e_northstar.npz", north, 5, 64, 128, seed=6)


if __name__ == "__main__":
    main()
