"""GPU suite (-m gpu): TrainEngine -- input validation, the two-stream step, data parallelism on a world-size-2 gloo
group (both ranks on cuda:0), and the Python objects around the native handles (copies, pickles, acc/depth losses)."""
import copy
import io
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import nerf_oracle as O
import parity_cases as P

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
CFG = dict(num_layers=4, hidden_size=128, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda", 0)


def _models(dev, seeds=(1, 2), cfg=CFG):
    import nerf_pytorch_amd as N
    mc, mf = N.FlexibleNeRFModel(**cfg), N.FlexibleNeRFModel(**cfg)
    mc.load_state_dict(O.init_params(cfg, seed=seeds[0]))
    mf.load_state_dict(O.init_params(cfg, seed=seeds[1]))
    return mc.to(dev), mf.to(dev)


def _rays(n, dev, seed=3):
    g = torch.Generator().manual_seed(seed)
    ro = torch.tensor([0.0, 0.0, 4.0]).expand(n, 3)
    rd = torch.randn(n, 3, generator=g) * 0.3
    rd[:, 2] = -1.0
    return O.pack_rays(ro, rd, 2.0, 6.0, rd).to(dev), torch.rand(n, 4, generator=g).to(dev)


def test_engine_validates_rays_and_target():
    """raw data_ptr()s must never be handed to the kernels unchecked."""
    import nerf_pytorch_amd as N
    dev = _dev()
    mc, mf = _models(dev)
    eng = N.TrainEngine(mc, mf, 16, 16, seed=5, world_size=1, rank=0)
    rays, rgba = _rays(64, dev)
    with pytest.raises(RuntimeError, match="float32"):
        eng.forward_backward(rays, rgba[:, :3].double())
    with pytest.raises(RuntimeError, match="must be a tensor on"):
        eng.forward_backward(rays.cpu(), rgba[:, :3])
    with pytest.raises(RuntimeError, match="contiguous rows"):
        eng.forward_backward(rays[:, :8], rgba[:, :3])
    with pytest.raises(RuntimeError, match="contiguous rows"):
        eng.forward_backward(torch.cat([rays, rays], 1)[:, :11], rgba[:, :3])
    with pytest.raises(RuntimeError, match="unit-stride"):
        eng.forward_backward(rays, rgba.t().contiguous().t()[:, :3])
    with pytest.raises(RuntimeError, match="one row of"):
        eng.forward_backward(rays, rgba[:32, :3])
    # the reference idiom target_s[..., :3] on an RGBA image: a (n, 3) view with row stride 4 -- must read the right pixels
    eng.forward_backward(rays, rgba[:, :3])
    torch.cuda.synchronize()
    g_view, l_view = eng.grad.clone(), eng.loss.clone()
    eng.forward_backward(rays, rgba[:, :3].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(g_view, eng.grad) and torch.equal(l_view, eng.loss)
    assert float(eng.loss[2]) > 0 and abs(float(eng.loss[0] + eng.loss[1]) - float(eng.loss[2])) < 1e-7


@pytest.mark.parametrize("precision,mode", [("fp32", None), ("f16x3_train", None), ("f16x3_train", "recompute"), ("fp32", "compact")])
def test_two_stream_step_equals_single_stream_step(precision, mode):
    """The overlapped graph (coarse backward on a side stream next to the fine pass) runs the same kernels on the same
    data: parameters after several steps are bit-identical to the single-stream order -- on the fp32 plans (two-stream by default up
    to 128 wide) and on the fp16-piece plans (two-stream by default at every width), dense and with the compacted / recomputing backward."""
    import nerf_pytorch_amd as N
    dev = _dev()
    out = []
    wide = dict(CFG, hidden_size=256)   # (the default depends on the arithmetic only above 128 wide)
    for overlap in (True, False, None):
        mc, mf = _models(dev, cfg=wide)
        if precision != "fp32":
            mc.set_training_precision(precision), mf.set_training_precision(precision)
        eng = N.TrainEngine(mc, mf, 32, 32, noise_std=0.2, seed=11, world_size=1, rank=0, overlap=overlap, backward=mode)
        assert eng.overlap == (overlap if overlap is not None else precision != "fp32")   # (the 8 x 256 default: by arithmetic)
        rays, rgba = _rays(640, dev)
        losses = [eng.step(rays, rgba[:, :3], ray_offset=0).clone() for _ in range(4)]
        torch.cuda.synchronize()
        out.append((mc.flat_params.clone(), mf.flat_params.clone(), torch.stack(losses)))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][2], out[1][2])
    assert float(out[0][2][-1, 2]) < float(out[0][2][0, 2])


@pytest.mark.parametrize("precision", ["f16x3_fwd", "f16x3_fwd_dgrad", "f16x3_train"])
def test_engine_on_fp16_piece_training_precisions_tracks_the_fp32_engine(precision):
    """set_training_precision (NERFHIP_PRECISION_F16X3_FWD / _FWD_DGRAD / _TRAIN; opt-in): the same engine, the same
    in-kernel draws, the forward (the data-gradient chain, the large weight-gradient blocks) on fp16 pieces.  One forward/backward:
    the loss agrees to 1e-4 relative and the flat gradient points the same way (cosine > 0.999, norm within 1 %) -- the element-wise
    bounds are tests/parity_cases.py's, on filtered rows; several optimizer steps: the loss falls as the fp32 engine's does.  A
    state_dict round trip is unaffected; round 3's bf16-piece names are refused."""
    import nerf_pytorch_amd as N
    dev = _dev()
    res = {}
    for prec in ("fp32", precision):
        mc, mf = _models(dev)
        if prec != "fp32":
            mc.set_training_precision(prec)
            mf.set_training_precision(prec)
            assert mc.training_precision == prec and set(mc.state_dict()) == set(O.init_params(CFG, seed=1))
        eng = N.TrainEngine(mc, mf, 32, 32, noise_std=0.2, seed=11, world_size=1, rank=0)
        rays, rgba = _rays(640, dev)
        eng.forward_backward(rays, rgba[:, :3], ray_offset=0)
        torch.cuda.synchronize()
        g0, l0 = eng.grad.clone(), eng.loss.clone()
        losses = [eng.step(rays, rgba[:, :3], ray_offset=0).clone() for _ in range(6)]
        torch.cuda.synchronize()
        res[prec] = (g0, l0, torch.stack(losses))
    (ga, la, sa), (gb, lb, sb) = res["fp32"], res[precision]
    assert abs(float(lb[2]) - float(la[2])) <= 1e-4 * abs(float(la[2])), (la, lb)
    cos = float(torch.dot(ga, gb) / (ga.norm() * gb.norm()))
    assert cos > 0.999 and abs(float(gb.norm() / ga.norm()) - 1.0) < 0.01, (cos, float(ga.norm()), float(gb.norm()))
    assert float(sb[-1, 2]) < float(sb[0, 2]) and abs(float(sb[-1, 2]) - float(sa[-1, 2])) < 0.02 * float(sa[0, 2]), (sa[:, 2], sb[:, 2])
    for gone in ("bf16", "bf16x3_train"):
        with pytest.raises(ValueError):
            mc.set_training_precision(gone)


@pytest.mark.parametrize("precision", ["fp32", "f16x3_train"])
def test_engine_backward_modes_compacted_recomputed_auto_track_the_dense_engine(precision):
    """TrainEngine(backward=...) (round 6): the compacted backward, the recomputing one and "auto" against the dense engine -- same
    weights, rays, in-kernel draws.  One forward/backward: the SAME loss bit for bit (the forward is the same kernel with or without
    its stash stores) and the flat gradient within 1e-5 of max|g| (another association of the same fp32 sums, zero terms dropped);
    the library's kept counts are below the sample counts (torch's init at sigma noise 0.2: ~40 % of the rows are zero).  120 steps on
    an empty white scene (every target pixel white under white_background: the nets learn sigma -> off, the rows die out): "auto"
    probes with its first step, follows the fraction to the compacted (fp16 pieces: then the recomputing) backward, and every mode's loss curve ends where
    the dense engine's does."""
    import nerf_pytorch_amd as N
    dev = _dev()
    rays, rgba = _rays(1024, dev)
    white = torch.ones(1024, 3, device=dev)
    res = {}
    for mode in ("dense", "compact", "recompute", "auto"):
        mc, mf = _models(dev)
        if precision != "fp32":
            mc.set_training_precision(precision)
            mf.set_training_precision(precision)
        eng = N.TrainEngine(mc, mf, 32, 32, noise_std=0.2, white_background=True, lr=1e-3, seed=11, world_size=1, rank=0, backward=mode)
        if mode != "auto":
            eng.forward_backward(rays, rgba[:, :3], ray_offset=0)
            torch.cuda.synchronize()
            g0, l0 = eng.grad.clone(), eng.loss.clone()
            kept = eng.backward_sample_counts()
        else:
            g0 = l0 = kept = None
        losses = torch.stack([eng.step(rays, white, ray_offset=0).clone() for _ in range(120)])
        torch.cuda.synchronize()
        res[mode] = (g0, l0, kept, losses, eng)
    gd, ld, _, sd, _ = res["dense"]
    for mode in ("compact", "recompute"):
        g, l, kept, s, _ = res[mode]
        assert torch.equal(l, ld), (mode, l, ld)
        assert float((g - gd).abs().max()) <= 1e-5 * float(gd.abs().max()), (mode, float((g - gd).abs().max()), float(gd.abs().max()))
        for name, total in (("coarse", 1024 * 32), ("fine", 1024 * 64)):
            assert kept[name][1] == total and 0 < kept[name][0] < total, (mode, name, kept)
    tail = lambda s: float(s[-20:, 2].mean())  # noqa: E731
    assert tail(sd) < 0.25 * float(sd[0, 2])           # (the scene is learned)
    for mode in ("compact", "recompute", "auto"):
        assert torch.isfinite(res[mode][3]).all()
        assert abs(tail(res[mode][3]) - tail(sd)) <= 0.05 * tail(sd) + 1e-4, (mode, tail(res[mode][3]), tail(sd))
    used = res["auto"][4].backward_modes_used
    assert sum(used["fine"]) == 120 and used["fine"][1] + used["fine"][2] > 20, used  # (the first step probes; then by the fraction)
    if precision != "fp32":
        assert used["fine"][2] + used["coarse"][2] > 0, used                           # (fp16 pieces: the recomputing mode was reached)
    else:
        assert used["fine"][2] == 0 and used["coarse"][2] == 0, used                   # (fp32: it never pays)


def test_engine_on_64_wide_nets_runs_the_fused_backward_by_default():
    """config/fern.yml's 4 x 64 nets (round 6, csrc/mlp64r.hip): a model whose plan has the LDS-resident image takes the fused one-kernel
    backward by default (over the register-image stash, mode 5); TrainEngine(backward="dense" / "fused" / "fused_compact" / "fused_stash" / "auto") on the same weights, rays and in-kernel
    draws: the SAME loss bit for bit (the stash-free resident forward computes what the stash-writing one does), flat gradients within
    1e-5 of max|g| of the dense engine's (another association of the same fp32 sums); 120 steps on an empty white scene end where the
    dense engine's do; "auto" runs fused over every sample until the list is known to drop rows, then fused over the list."""
    import nerf_pytorch_amd as N
    dev = _dev()
    cfg = dict(num_layers=4, hidden_size=64, skip_connect_every=3, num_encoding_fn_xyz=6, num_encoding_fn_dir=4)
    rays, rgba = _rays(1024, dev)
    white = torch.ones(1024, 3, device=dev)
    mc, mf = _models(dev, cfg=cfg)
    assert mc.fused_backward_available() == 5 and mc.backward_compaction == 5 and mf.backward_compaction == 5
    wide = _models(dev)[0]
    assert not wide.fused_backward_available() and wide.backward_compaction == 0
    with pytest.raises(Exception, match="fused backward"):
        wide.set_backward_compaction("fused")
    res = {}
    for mode in ("dense", None, "fused", "fused_compact", "fused_stash", "auto"):
        mc, mf = _models(dev, cfg=cfg)
        eng = N.TrainEngine(mc, mf, 32, 32, noise_std=0.2, white_background=True, lr=1e-3, seed=11, world_size=1, rank=0, backward=mode)
        if mode != "auto":
            eng.forward_backward(rays, rgba[:, :3], ray_offset=0)
            torch.cuda.synchronize()
            g0, l0 = eng.grad.clone(), eng.loss.clone()
        else:
            g0 = l0 = None
        losses = torch.stack([eng.step(rays, white, ray_offset=0).clone() for _ in range(120)])
        torch.cuda.synchronize()
        res[mode] = (g0, l0, losses, eng)
    gd, ld, sd, _ = res["dense"]
    assert res[None][3].mc.backward_compaction == 5                       # (backward=None: the models' default)
    assert torch.equal(res[None][0], res["fused_stash"][0])
    assert torch.equal(res["fused"][0], res["fused_stash"][0])            # (the same arithmetic on the same values: bit-identical)
    for mode in ("fused", "fused_compact", "fused_stash"):
        g, l, s, eng = res[mode]
        assert torch.equal(l, ld), (mode, l, ld)
        assert float((g - gd).abs().max()) <= 1e-5 * float(gd.abs().max()), (mode, float((g - gd).abs().max()), float(gd.abs().max()))
    kept = res["fused_compact"][3].backward_sample_counts()
    assert kept["coarse"][1] == 1024 * 32 and kept["fine"][1] == 1024 * 64 and res["fused"][3].backward_sample_counts()["fine"] is None
    tail = lambda s: float(s[-20:, 2].mean())  # noqa: E731
    assert tail(sd) < 0.25 * float(sd[0, 2])
    for mode in (None, "fused", "fused_compact", "fused_stash", "auto"):
        assert torch.isfinite(res[mode][2]).all()
        assert abs(tail(res[mode][2]) - tail(sd)) <= 0.05 * tail(sd) + 1e-4, (mode, tail(res[mode][2]), tail(sd))
    used = res["auto"][3].backward_modes_used
    assert sum(used["fine"]) == 120 and used["fine"][0] + used["fine"][1] + used["fine"][2] == 0 and used["fine"][4] > 20, used


def test_engine_fed_external_draws_equals_in_kernel_draws(gpu):
    """TrainEngine.step(draws=...) (the PSNR experiment's "engine on torch's draws" arm): feeding the engine the numbers
    nerfhip_rng_fill reports for (seed, stream, element) reproduces the in-kernel Philox step bit for bit."""
    import nerf_pytorch_amd as N
    dev = _dev()
    n, nc, nf, seed = 200, 32, 48, 4242
    rays, rgba = _rays(n, dev)
    res = []
    for external in (False, True):
        mc, mf = _models(dev)
        eng = N.TrainEngine(mc, mf, nc, nf, noise_std=0.3, seed=seed, world_size=1, rank=0, overlap=False)
        draws = None
        if external:
            f = lambda kind, stream, cnt: torch.from_numpy(gpu.rng_fill(kind, seed, stream, 0, n * cnt).reshape(n, cnt)).to(dev)  # noqa: E731
            draws = (f(0, 0, nc), f(1, 1, nc), f(0, 2, nf), f(1, 3, nc + nf))
        eng.forward_backward(rays, rgba[:, :3], draws=draws)
        torch.cuda.synchronize()
        res.append((eng.grad.clone(), eng.loss.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    with pytest.raises(RuntimeError, match="random-draw tensor"):
        eng.forward_backward(rays, rgba[:, :3], draws=(torch.rand(n, nc + 1, device=dev), None, None, None))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("overlap", [1, 0])
def test_two_rank_engine_stays_in_lockstep_and_matches_one_process(tmp_path, overlap):
    """two ranks (gloo, both on cuda:0), each its own shard of every step's rays: the ranks' weights stay
    bit-identical, and equal a one-process run on the concatenated batch up to fp32 summation order."""
    dev = _dev()
    sys.path.insert(0, HERE)
    import dp_worker as D
    steps, n = 5, 256
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dp_worker.py"), str(tmp_path), str(steps), str(n),
                                       str(overlap)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    for k in ("grad0", "pc", "pf"):
        assert np.array_equal(r0[k], r1[k]), "ranks diverged: " + k
    assert r0["coll"].shape == (2,) and np.isfinite(r0["coll"]).all()   # (collective_times_ms ran on both ranks and left the gradient alone)
    image, pose = D.scene(dev)
    _, _, eng = D.make_engine(dev, 1, 0, bool(overlap))
    grad0, pc, pf, _ = D.run(eng, image, pose, steps, 2 * n, dev)
    scale = float(np.abs(grad0).max())
    assert float(np.abs(r0["grad0"] - grad0).max()) <= 2e-5 * scale, float(np.abs(r0["grad0"] - grad0).max()) / scale
    # Adam divides by sqrt(v): an entry whose gradient is at the round-off floor may step the other way, so compare the
    # bulk tightly and bound the tail by what `steps` sign flips could move
    for a, b in ((r0["pc"], pc), (r0["pf"], pf)):
        d = np.abs(a - b)
        assert float(np.quantile(d, 0.999)) <= 2e-4, float(np.quantile(d, 0.999))
        assert float(d.max()) <= 2.1 * 5e-3 * steps


@pytest.mark.parametrize("overlap", [1, 0])
def test_rccl_branch_one_rank_group_equals_plain_engine(overlap):
    """the nccl (RCCL) code path -- asynchronous work handles, stream-ordered wait() -- executes on
    hardware: a one-rank nccl group with always_reduce=True must reproduce the engine without a group bit for bit
    (tests/nccl_worker.py)."""
    _dev()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    p = subprocess.run([sys.executable, os.path.join(HERE, "nccl_worker.py"), str(overlap), "5", "384"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0, p.stdout.decode()[-3000:]
    assert "nccl one-rank engine ok" in p.stdout.decode()


def test_ray_sharded_eval_is_bit_identical_to_one_rank():
    """BASELINE config 5 (eval_nerf.py:158-190): every rank renders its own block of image rows with no collective
    (eval_utils.render_pose_rows); the blocks of a 2-, 3- and 8-way split, in rank order, are the single-rank image bit
    for bit -- colour, disparity (NaN pixels included) and accumulation, coarse and fine."""
    import nerf_pytorch_amd as N
    dev = _dev()
    mc, mf = _models(dev, seeds=(7, 8))
    H, W, focal = 37, 40, 50.0           # 37 rows: ragged shards
    pose = torch.eye(4)
    pose[2, 3] = 4.0
    pose = pose.to(dev)
    ex, ed = N.get_embedding_function(10, True, True), N.get_embedding_function(4, True, True)
    opts = N.make_options(32, 32, perturb=False, radiance_field_noise_std=0.0, chunksize=4096)
    with torch.no_grad():
        ro, rd = N.get_ray_bundle(H, W, focal, pose[:3, :4])
        ref = N.run_one_iter_of_nerf(H, W, focal, mc, mf, ro, rd, opts, mode="validation", encode_position_fn=ex,
                                     encode_direction_fn=ed)
        one, (lo, hi) = N.render_pose_rows(H, W, focal, pose, mc, mf, opts, ex, ed)
        assert (lo, hi) == (0, H)
        for a, b in zip(one, ref):
            assert torch.equal(torch.nan_to_num(a, nan=-1.0), torch.nan_to_num(b, nan=-1.0))
        for world in (2, 3, 8):
            parts = [N.render_pose_rows(H, W, focal, pose, mc, mf, opts, ex, ed, rank=r, world_size=world) for r in range(world)]
            assert [p[1] for p in parts] == [N.parallel.shard_bounds(H, r, world) for r in range(world)]
            for k in range(6):
                got = torch.cat([p[0][k] for p in parts], dim=0)
                assert got.shape == ref[k].shape
                assert torch.equal(torch.nan_to_num(got, nan=-1.0), torch.nan_to_num(ref[k], nan=-1.0)), (world, k)
                assert torch.equal(torch.isnan(got), torch.isnan(ref[k]))


def test_model_deepcopy_pickle_and_device_moves_own_their_native_plans():
    """copy.deepcopy / torch.save of a model must not duplicate the native plan pointer."""
    import nerf_pytorch_amd as N
    dev = _dev()
    m, _ = _models(dev)
    x = torch.randn(300, m.dim_xyz + m.dim_dir, device=dev)
    with torch.no_grad():
        y = m(x)
    c = copy.deepcopy(m)
    assert c._plan != m._plan and c.flat_params.data_ptr() != m.flat_params.data_ptr()
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    r = torch.load(buf, weights_only=False)
    assert r._plan not in (m._plan, c._plan)
    with torch.no_grad():
        assert torch.equal(c(x), y) and torch.equal(r(x), y)
        # the copies are independent: changing one does not touch the others
        c.layer1.weight.mul_(0.5)
        assert torch.equal(m(x), y) and not torch.equal(c(x), y)
    for name, p in c.named_parameters():  # still views of ONE flat buffer
        assert p.data_ptr() >= c.flat_params.data_ptr() and p.data_ptr() < c.flat_params.data_ptr() + 4 * c.num_flat_params
    del m, c, r  # three plans, three destroys
    import gc
    gc.collect()
    torch.cuda.synchronize()


def test_fused_path_propagates_accumulation_and_depth_losses():
    """a loss with an acc / disparity term must reach the parameters on the fused path exactly as on the
    generic composition (which mirrors the reference op by op)."""
    import nerf_pytorch_amd as N
    dev = _dev()
    ex, ed = N.get_embedding_function(10, True, True), N.get_embedding_function(4, True, True)
    opts = N.make_options(24, 24, perturb=False, radiance_field_noise_std=0.0)
    g = torch.Generator().manual_seed(12)
    ro = torch.tensor([0.0, 0.0, 4.0]).expand(40, 3).contiguous().to(dev)
    rd = torch.randn(40, 3, generator=g) * 0.3
    rd[:, 2] = -1.0
    rd = rd.to(dev)
    tgt = torch.rand(40, 3, generator=g).to(dev)
    grads = []
    for fused in (True, False):
        mc, mf = _models(dev, seeds=(5, 6))
        # a plain callable hides the FlexibleNeRFModel type: predict_and_render_radiance then composes the unit kernels
        a, b = (mc, mf) if fused else ((lambda t, m=mc: m(t)), (lambda t, m=mf: m(t)))
        out = N.run_one_iter_of_nerf(40, 1, 30.0, a, b, ro, rd, opts, encode_position_fn=ex, encode_direction_fn=ed)
        rgb_c, disp_c, acc_c, rgb_f, disp_f, acc_f = out
        loss = ((rgb_f - tgt) ** 2).mean() + 0.3 * ((acc_c - 0.5) ** 2).mean() + 0.2 * (acc_f ** 2).mean() \
            + 0.1 * torch.nan_to_num(disp_f).mean() + 0.05 * torch.nan_to_num(disp_c).mean()
        loss.backward()
        grads.append(torch.cat([p.grad.reshape(-1) for p in list(mc.parameters()) + list(mf.parameters())]).cpu().numpy())
    scale = float(np.abs(grads[1]).max())
    assert scale > 0
    P.close(grads[0], grads[1], 2e-5 * scale, what="acc/depth/disp loss gradient, fused vs generic")


def test_autograd_node_runs_its_backward_in_the_mode_of_its_forward():
    """The backward data flow is an option of the plan, and what a training forward leaves in its workspace depends on it (the general
    stash, nothing, the register-image stash of 64-wide nets: nerfhip_plan_set_bwd_compaction).  The autograd node of the fused render
    (run_one_iter_of_nerf -> loss.backward(), the reference's own loop: train_nerf.py:226-259) pins the modes its forward ran in: calling
    set_backward_compaction between the forward and the backward changes later forwards, not this node's gradient -- bit for bit --
    and the models keep what was asked for."""
    import nerf_pytorch_amd as N
    dev = _dev()
    cfg = dict(num_layers=4, hidden_size=64, skip_connect_every=3, num_encoding_fn_xyz=6, num_encoding_fn_dir=4)
    ex, ed = N.get_embedding_function(6, True, True), N.get_embedding_function(4, True, True)
    opts = N.make_options(24, 24, perturb=False, radiance_field_noise_std=0.0)
    g = torch.Generator().manual_seed(31)
    ro = torch.tensor([0.0, 0.0, 4.0]).expand(300, 3).contiguous().to(dev)
    rd = torch.randn(300, 3, generator=g) * 0.3
    rd[:, 2] = -1.0
    rd = rd.to(dev)
    tgt = torch.rand(300, 3, generator=g).to(dev)
    for first, then in ((None, False), (None, "fused"), (False, "fused_stash"), ("recompute", False), (True, "recompute")):
        grads = []
        for switch in (False, True):
            mc, mf = _models(dev, cfg=cfg, seeds=(5, 6))
            if first is not None:
                mc.set_backward_compaction(first), mf.set_backward_compaction(first)
            out = N.run_one_iter_of_nerf(300, 1, 30.0, mc, mf, ro, rd, opts, encode_position_fn=ex, encode_direction_fn=ed)
            loss = ((out[0] - tgt) ** 2).mean() + ((out[3] - tgt) ** 2).mean()
            if switch:
                mc.set_backward_compaction(then), mf.set_backward_compaction(then)
                want = mc.backward_compaction
            loss.backward()
            if switch:
                assert mc.backward_compaction == want and N._lib.get_lib().plan_bwd_compaction(mc._plan) == want
            grads.append(torch.cat([p.grad.reshape(-1) for p in list(mc.parameters()) + list(mf.parameters())]))
        assert torch.isfinite(grads[1]).all() and float(grads[0].abs().max()) > 0
        assert torch.equal(grads[0], grads[1]), (first, then, float((grads[0] - grads[1]).abs().max()))


@pytest.mark.parametrize("hidden", [64, 128])
def test_models_in_auto_mode_pick_their_backward_through_the_reference_loop(hidden):
    """set_backward_compaction("auto") on the models (no TrainEngine): the reference's own loop -- run_one_iter_of_nerf, loss.backward(),
    torch.optim.Adam (train_nerf.py:226-261) -- on a scene that empties out: the autograd node's backward reports {kept, total} per net
    (the two nets share one set of backward buffers there: the passes are issued one by one with the copy in between), the models
    move from their dense mode (0; 64-wide nets: fused over the stash, 5) to the list (1 / 4) once it is known to drop enough rows, and
    the gradient of the mode they end in equals the dense one on the same weights and rays."""
    import copy
    import nerf_pytorch_amd as N
    dev = _dev()
    cfg = dict(num_layers=4, hidden_size=hidden, skip_connect_every=3 if hidden == 64 else 4, num_encoding_fn_xyz=6, num_encoding_fn_dir=4)
    ex, ed = N.get_embedding_function(6, True, True), N.get_embedding_function(4, True, True)
    opts = N.make_options(32, 32, perturb=False, radiance_field_noise_std=0.0, white_background=True)
    g = torch.Generator().manual_seed(7)
    n = 512
    ro = torch.tensor([0.0, 0.0, 4.0]).expand(n, 3).contiguous().to(dev)
    rd = torch.randn(n, 3, generator=g) * 0.3
    rd[:, 2] = -1.0
    rd = rd.to(dev)
    white = torch.ones(n, 3, device=dev)
    mc, mf = _models(dev, cfg=cfg, seeds=(5, 6))
    dense_mode = 5 if hidden == 64 else 0
    assert mc.backward_compaction == dense_mode
    mc.set_backward_compaction("auto"), mf.set_backward_compaction("auto")
    assert mc.backward_compaction == dense_mode and mf.backward_compaction == dense_mode
    opt = torch.optim.Adam(list(mc.parameters()) + list(mf.parameters()), lr=1e-3)
    seen = set()
    for it in range(150):
        out = N.run_one_iter_of_nerf(n, 1, 30.0, mc, mf, ro, rd, opts, encode_position_fn=ex, encode_direction_fn=ed)
        seen.add((mc.backward_compaction, mf.backward_compaction))
        loss = ((out[0] - white) ** 2).mean() + ((out[3] - white) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        if it % 10 == 9:
            torch.cuda.synchronize()   # (lets the asynchronous statistics land: a real loop's logging does the same)
    assert torch.isfinite(loss).all() and float(loss.detach()) < 0.05, float(loss.detach())
    list_mode = 4 if hidden == 64 else 1
    assert mc._auto_frac is not None and mf._auto_frac is not None and mf._auto_frac > 0.5, (mc._auto_frac, mf._auto_frac)
    assert any(m[1] == list_mode for m in seen) and mf.backward_compaction == list_mode, (seen, mf.backward_compaction)
    # the gradient in the mode the loop ended in, against the dense backward of the same weights
    grads = []
    for dense in (False, True):
        a, b = copy.deepcopy(mc), copy.deepcopy(mf)
        if dense:
            a.set_backward_compaction(False), b.set_backward_compaction(False)
        else:
            a.set_backward_compaction({1: True, 4: "fused_compact"}.get(mc.backward_compaction, False))
            b.set_backward_compaction({1: True, 4: "fused_compact"}[mf.backward_compaction])
        out = N.run_one_iter_of_nerf(n, 1, 30.0, a, b, ro, rd, opts, encode_position_fn=ex, encode_direction_fn=ed)
        (((out[0] - 0.5) ** 2).mean() + ((out[3] - 0.5) ** 2).mean()).backward()
        grads.append(torch.cat([p.grad.reshape(-1) for p in list(a.parameters()) + list(b.parameters())]))
    assert float((grads[0] - grads[1]).abs().max()) <= 1e-5 * float(grads[1].abs().max()) + 1e-12
