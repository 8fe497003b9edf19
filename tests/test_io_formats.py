"""Checkpoint / cached-dataset formats and the PNG writer: host logic, runs without a GPU."""
import io
import os
import types
import zlib

import numpy as np
import pytest
import torch

import nerf_pytorch_amd as N
from nerf_pytorch_amd import io_utils as IO
from nerf_pytorch_amd.eval_utils import png_bytes

REF_CKPT = "/root/reference/pretrained/lego-lowres/checkpoint199999.ckpt"
CFG = dict(num_layers=4, hidden_size=128, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)


def fake_engine(mc, mf):
    """The slice of TrainEngine the converters touch, on CPU tensors (TrainEngine itself needs a GPU)."""
    tot = mc.num_flat_params + (mf.num_flat_params if mf is not None else 0)
    return types.SimpleNamespace(mc=mc, mf=mf, exp_avg=torch.zeros(tot), exp_avg_sq=torch.zeros(tot), step_count=0,
                                 lr=5e-3, betas=(0.9, 0.999), eps=1e-8, repack=lambda: None)


def test_parameters_follow_the_flat_order():
    m = N.FlexibleNeRFModel(**CFG)
    ps, flat = list(m.parameters()), m.flat_params
    off = 0
    for p, q in zip(ps, m._ordered_params()):
        assert p is q and p.data_ptr() == flat.data_ptr() + 4 * off
        off += p.numel()
    assert off == flat.numel()


def test_checkpoint_round_trip(tmp_path):
    torch.manual_seed(3)
    mc, mf = N.FlexibleNeRFModel(**CFG), N.FlexibleNeRFModel(**CFG)
    eng = fake_engine(mc, mf)
    eng.exp_avg.normal_()
    eng.exp_avg_sq.uniform_()
    eng.step_count, eng.lr = 1234, 1.5e-3
    sd = IO.engine_optimizer_state_dict(eng)
    # the dict is what torch.optim.Adam itself accepts for these parameters (the reference's resume path,
    # train_nerf.py:156-163)
    opt = torch.optim.Adam(list(mc.parameters()) + list(mf.parameters()), lr=5e-3)
    opt.load_state_dict(sd)
    st = opt.state[list(mc.parameters())[0]]
    assert int(st["step"]) == 1234 and torch.equal(st["exp_avg"].reshape(-1), eng.exp_avg[:st["exp_avg"].numel()])
    assert opt.param_groups[0]["lr"] == 1.5e-3
    path = str(tmp_path / "checkpoint01234.ckpt")
    IO.save_checkpoint(path, 1234, mc, mf, sd, torch.tensor(0.01), 20.0, height=100, width=100, focal_length=138.9)
    ck = torch.load(path, weights_only=False)
    assert set(ck) == {"iter", "model_coarse_state_dict", "model_fine_state_dict", "optimizer_state_dict", "loss", "psnr",
                       "height", "width", "focal_length"}
    mc2, mf2 = N.FlexibleNeRFModel(**CFG), N.FlexibleNeRFModel(**CFG)
    eng2 = fake_engine(mc2, mf2)
    ck = IO.load_checkpoint(path, mc2, mf2, engine=eng2)
    assert ck["iter"] == 1234 and eng2.step_count == 1234 and eng2.lr == 1.5e-3
    assert torch.equal(mc2.flat_params, mc.flat_params) and torch.equal(mf2.flat_params, mf.flat_params)
    assert torch.equal(eng2.exp_avg, eng.exp_avg) and torch.equal(eng2.exp_avg_sq, eng.exp_avg_sq)
    # coarse-only checkpoints store None for the fine net (train_nerf.py:376-378)
    IO.save_checkpoint(path, 1, mc, None, None, 0.0, 0.0)
    assert torch.load(path, weights_only=False)["model_fine_state_dict"] is None
    IO.load_checkpoint(path, mc2, None)


@pytest.mark.skipif(not os.path.exists(REF_CKPT), reason="reference tree not present (build container only)")
def test_reads_the_references_own_checkpoint():
    """The reference's pretrained file (torch 1.x optimizer layout: id() keys, integer step)."""
    mc, mf = N.FlexibleNeRFModel(**CFG), N.FlexibleNeRFModel(**CFG)
    eng = fake_engine(mc, mf)
    ck = IO.load_checkpoint(REF_CKPT, mc, mf, engine=eng, map_location="cpu")
    assert ck["iter"] == 199999 and eng.step_count == 200000
    assert abs(eng.lr - 0.0007924538949670465) < 1e-12
    assert torch.equal(mc.layer1.weight, ck["model_coarse_state_dict"]["layer1.weight"])
    osd = ck["optimizer_state_dict"]
    keys = osd["param_groups"][0]["params"]
    first, last = osd["state"][keys[0]], osd["state"][keys[-1]]
    assert torch.equal(eng.exp_avg[:first["exp_avg"].numel()], first["exp_avg"].reshape(-1))
    assert torch.equal(eng.exp_avg_sq[-last["exp_avg_sq"].numel():], last["exp_avg_sq"].reshape(-1))
    assert float(eng.exp_avg_sq.min()) >= 0


def test_cached_example_round_trip(tmp_path):
    g = torch.Generator().manual_seed(0)
    bundle, tgt = torch.randn(2, 6, 5, 3, generator=g), torch.rand(6, 5, 4, generator=g)
    p = str(tmp_path / "0003.data")
    IO.save_cached_example(p, 6, 5, 7.5, tgt, ray_bundle=bundle)
    d = IO.load_cached_example(p)
    assert set(d) == {"height", "width", "focal_length", "ray_bundle", "target"}      # cache_dataset.py:104-110
    assert torch.equal(d["ray_bundle"], bundle) and d["focal_length"] == 7.5
    IO.save_cached_example(p, 6, 5, 7.5, tgt, ray_origins=bundle[0], ray_directions=bundle[1])
    d = IO.load_cached_example(p)
    assert set(d) == {"height", "width", "focal_length", "ray_origins", "ray_directions", "target"}  # :124-131
    assert torch.equal(d["ray_directions"], bundle[1])


def test_png_writer():
    rng = np.random.RandomState(0)
    for shape in ((7, 5, 3), (4, 9)):
        img = rng.randint(0, 256, size=shape).astype(np.uint8)
        data = png_bytes(img)
        assert data[:8] == b"\x89PNG\r\n\x1a\n"
        # decode by hand: IHDR + one IDAT of filter-0 scanlines
        w, h = int.from_bytes(data[16:20], "big"), int.from_bytes(data[20:24], "big")
        assert (h, w) == shape[:2]
        n = int.from_bytes(data[33:37], "big")
        assert data[37:41] == b"IDAT"
        raw = np.frombuffer(zlib.decompress(data[41:41 + n]), np.uint8).reshape(h, -1)
        assert np.all(raw[:, 0] == 0) and np.array_equal(raw[:, 1:].reshape(shape), img)
        PIL = pytest.importorskip("PIL.Image")
        assert np.array_equal(np.array(PIL.open(io.BytesIO(data))), img)
