"""CPU suite: the product library loads and exports every symbol include/nerfhip.h declares; host-only plan / packing
logic; the Python package's host behaviour (state_dict compatibility, loud failure without a GPU); the N>1 data-parallel
contract with gloo, world_size 2."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

import nerf_oracle as O
from conftest import ROOT, gold

sys.path.insert(0, ROOT)
import nerf_pytorch_amd as N  # noqa: E402
from nerf_pytorch_amd import _lib as L  # noqa: E402


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(L.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "nerf-pytorch_amd", "csrc"), "lib", "-j8"], check=True)
    return L.get_lib()


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "nerfhip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(nerfhip_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    raw = C.CDLL(L.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), "libnerfhip.so does not export " + name
    assert declared == set(L.EXPORTED_SYMBOLS), declared ^ set(L.EXPORTED_SYMBOLS)
    assert lib.is_emulated() == 0 and 100 <= lib.version() < L.DIAG_FLAG  # (a `make variant` build reports version + DIAG_FLAG)


def test_errors_are_reported_not_thrown(lib):
    with pytest.raises(L.NerfHipError, match="bad arguments"):
        lib.cumprod_exclusive(None, 1, 4, None, None)
    bad = L.ModelCfg(4, 600, 4, 10, 4, 1, 1, 1, 1, 1)
    assert not lib.plan_create(C.byref(bad))
    assert b"hidden_size" in lib.last_error()
    bad = L.ModelCfg(4, 128, 4, 17, 4, 1, 1, 1, 1, 1)
    assert not lib.plan_create(C.byref(bad))


@pytest.mark.parametrize("geo", [(8, 256, 4, 10, 4, True), (4, 128, 4, 10, 4, True), (8, 128, 3, 6, 4, True),
                                 (4, 128, 4, 10, 4, False), (6, 256, 2, 10, 4, True), (4, 128, 4, 12, 4, True),
                                 (8, 256, 4, 16, 6, True), (3, 64, 1, 11, 0, False), (2, 512, 4, 12, 10, True)])
def test_plan_layout_and_pack_table(lib, geo):
    Lr, W, sk, lx, ld, view = geo
    cfg = dict(num_layers=Lr, hidden_size=W, skip_connect_every=sk, num_encoding_fn_xyz=lx, num_encoding_fn_dir=ld,
               use_viewdirs=view)
    mc = L.ModelCfg(Lr, W, sk, lx, ld, 1, 1, 1, 1, int(view))
    plan = lib.plan_create(C.byref(mc))
    assert plan
    shapes = O.param_shapes(cfg)
    assert lib.plan_num_tensors(plan) == len(shapes)
    off = 0
    for i, (name, shape) in enumerate(shapes):
        nm, o, r, c = C.c_char_p(), C.c_int64(), C.c_int(), C.c_int()
        lib.plan_tensor_info(plan, i, C.byref(nm), C.byref(o), C.byref(r), C.byref(c))
        assert nm.value.decode() == name and o.value == off
        assert (r.value, c.value) == (shape[0], shape[1] if len(shape) == 2 else 0)
        off += int(np.prod(shape))
    assert lib.plan_num_params(plan) == off
    n = lib.plan_packed_floats(plan)
    table = np.empty(n, np.int32)
    lib.plan_pack_index(plan, table.ctypes.data)
    assert table.min() >= -1 and table.max() < off
    # every parameter is gathered into the forward image exactly once
    dx, dd = O.model_dims(cfg)
    counts = np.bincount(table[table >= 0], minlength=off)
    assert counts.min() >= 1
    if (Lr, W, lx, ld) == (8, 256, 10, 4) and view:
        assert off == 595844
    lib.plan_destroy(plan)


def test_thin_weight_blocks_ride_as_side_tiles(lib):
    """wgrad.hip side tiles: fc_alpha and the direction columns have no weight-gradient job of their own; 512-wide nets
    (half-region jobs) and 64-wide nets (no job of 8 tiles) keep them."""
    def describe(*geo):
        plan = lib.plan_create(C.byref(L.ModelCfg(*geo, 1, 1, 1, 1, 1)))
        assert plan
        buf = C.create_string_buffer(1 << 14)
        lib.plan_describe(plan, buf, len(buf))
        lib.plan_destroy(plan)
        lines = buf.value.decode().splitlines()
        jobs = [dict(zip(l.split()[0::2], l.split()[1::2])) for l in lines[1:]]
        return lines[0], jobs
    head, jobs = describe(8, 256, 4, 10, 4)
    assert "kernel_width 256" in head and len(jobs) == 12                       # 14 weight blocks, two of them side tiles
    assert sorted((j["side"], j["side_tiles"]) for j in jobs if j["side"] != "0") == [("1", "1"), ("2", "1")]
    head, jobs = describe(4, 128, 4, 10, 4)
    assert "kernel_width 128" in head and len(jobs) == 7 and sum(j["side"] != "0" for j in jobs) == 2
    head, jobs = describe(8, 128, 4, 10, 4)                                      # the skip layer's 64-row block does not fit the 4-wave stage
    assert len(jobs) == 12 and sum(j["side"] != "0" for j in jobs) == 2
    for geo in ((4, 64, 3, 6, 4), (3, 512, 2, 10, 4)):
        assert all(j["side"] == "0" for j in describe(*geo)[1])


def test_split_precision_plans_guests_and_image_geometry(lib):
    """Level-4 plans: the large weight-gradient blocks leave the fp32 job list and take the thin blocks that share a region with
    them along as guests (wgrad_f16.hip SA / SB) -- what stays are layer1's block and fc_rgb's; fp16-piece plans carry their images
    in the geometry of the two-waves-per-SIMD kernels (mlp_f16w.hip).  The values that named round 3's bf16-piece plans are refused."""
    def describe(prec, *geo):
        plan = lib.plan_create_ex(C.byref(L.ModelCfg(*geo, 1, 1, 1, 1, 1)), prec)
        assert plan
        buf = C.create_string_buffer(1 << 14)
        lib.plan_describe(plan, buf, len(buf))
        lib.plan_destroy(plan)
        lines = buf.value.decode().splitlines()
        return lines[0], lines[1:]
    head, jobs = describe(8, 8, 256, 4, 10, 4)                  # NERFHIP_PRECISION_F16X3_TRAIN
    assert head.split()[-2:] == ["two_wave_images", "1"] and len(jobs) == 2, (head, jobs)
    head, jobs = describe(8, 4, 128, 4, 10, 4)                  # 128-wide: layers_dir's two blocks stay fp32 jobs (one a side tile)
    assert len(jobs) == 3, (head, jobs)
    head, jobs = describe(7, 8, 256, 4, 10, 4)                  # _F16X3_FWD_DGRAD: every weight-gradient block on the fp32 kernel
    assert head.split()[-1] == "1" and len(jobs) == 12
    head, jobs = describe(5, 8, 256, 4, 10, 4)                  # _F16X3: inference-only
    assert head.split()[-1] == "1"
    assert describe(0, 8, 256, 4, 10, 4)[0].split()[-1] == "0"  # fp32 plans carry no fp16-piece image
    for removed in (1, 2, 3, 4):
        assert not lib.plan_create_ex(C.byref(L.ModelCfg(8, 256, 4, 10, 4, 1, 1, 1, 1, 1)), removed)
        assert b"removed in round 5" in lib.last_error()


def test_model_state_dict_is_reference_compatible(lib):
    w = gold("lego_lowres_weights.npz")
    m = N.FlexibleNeRFModel(num_layers=4, hidden_size=128, skip_connect_every=4, num_encoding_fn_xyz=10,
                            num_encoding_fn_dir=4)
    ref = {k[2:]: torch.from_numpy(w[k]) for k in w.files if k.startswith("c_")}
    assert list(m.state_dict().keys()) == list(ref.keys())
    m.load_state_dict(ref)
    for k, v in m.state_dict().items():
        assert torch.equal(v, ref[k])
    # parameters alias one flat buffer in state_dict order
    flat = m.flat_params
    assert flat.numel() == 84548
    assert torch.equal(flat[:128 * 63].view(128, 63), ref["layer1.weight"])
    with torch.no_grad():
        m.layer1.bias.add_(1.0)
    assert torch.equal(flat[128 * 63:128 * 63 + 128], ref["layer1.bias"] + 1.0)
    # same construction order as the reference => same init under the same seed
    torch.manual_seed(42)
    a = N.FlexibleNeRFModel(8, 256, 4, 10, 4)
    torch.manual_seed(42)
    lin = torch.nn.Linear(63, 256)
    assert torch.equal(a.layer1.weight, lin.weight)


def test_pure_torch_helpers_match_the_reference_contract():
    """meshgrid_xy (nerf/nerf_helpers.py:28-40), mse2psnr / img2mse (:9-17): the three exports without a kernel."""
    ii, jj = N.meshgrid_xy(torch.arange(3), torch.arange(2))          # numpy "xy": shapes (len(t2), len(t1))
    assert ii.tolist() == [[0, 1, 2], [0, 1, 2]] and jj.tolist() == [[0, 0, 0], [1, 1, 1]]
    oi, oj = O.meshgrid_xy(torch.arange(5.0), torch.arange(7.0)) if hasattr(O, "meshgrid_xy") else \
        [t.transpose(-1, -2) for t in torch.meshgrid(torch.arange(5.0), torch.arange(7.0), indexing="ij")]
    gi, gj = N.meshgrid_xy(torch.arange(5.0), torch.arange(7.0))
    assert torch.equal(gi, oi) and torch.equal(gj, oj)
    assert N.mse2psnr(0.01) == pytest.approx(20.0) and N.mse2psnr(0) == pytest.approx(50.0)     # KAT7
    a, b = torch.rand(11, 3), torch.rand(11, 3)
    assert torch.equal(N.img2mse(a, b), torch.nn.functional.mse_loss(a, b))


def test_no_cpu_fallback(lib):
    m = N.FlexibleNeRFModel()
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros(4, m.dim_xyz + m.dim_dir))
    with pytest.raises(RuntimeError, match="no CPU path"):
        N.positional_encoding(torch.zeros(4, 3))
    with pytest.raises(RuntimeError, match="no CPU path"):
        N.volume_render_radiance_field(torch.zeros(2, 4, 4), torch.zeros(2, 4), torch.zeros(2, 3))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "nerf-pytorch_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "nerf_oracle" not in src and "oracle" not in src.replace("the oracle's bits", ""), fn
            assert "libnerfhip_emu" not in src, fn


def test_product_reads_no_environment_and_ships_one_kernel_set():
    """no developer knobs in the shipped library -- neither the Python package nor the C/HIP sources
    read the environment; the instrumentation (timelines, phase stamps) only exists behind the `make dbg` macros; the product's
    compiler flags define NO NH* switch at all (A/B schedules and wrong-result cost-attribution switches are `make variant` builds,
    compiled with -DNH_DIAG, marked in nerfhip_version() and refused by get_lib()); a wrong-result switch without NH_DIAG does not
    compile (csrc/nh_diag.h); and ONE kernel set ships: fp32 + fp16 pieces (round 3's bf16-piece family is gone)."""
    import re
    pkg = os.path.join(ROOT, "nerf-pytorch_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "os.environ" not in src and "getenv" not in src, fn
    csrc = os.path.join(pkg, "csrc")
    exp = set()
    for fn in os.listdir(csrc):
        if fn.endswith((".hip", ".cpp", ".h")):
            src = open(os.path.join(csrc, fn)).read()
            assert "getenv" not in src, fn
            if fn != "nh_diag.h":
                exp |= set(re.findall(r"\bNH[A-Z0-9]*_EXP_[A-Z0-9_]+", src))
                if re.search(r"\bNH[A-Z0-9]*_EXP_", src):
                    assert '#include "nh_diag.h"' in src, fn   # (every file with a wrong-result switch sits behind the fence)
    fence = open(os.path.join(csrc, "nh_diag.h")).read()
    assert exp and all("defined(%s)" % m in fence for m in exp), sorted(m for m in exp if "defined(%s)" % m not in fence)
    mk = open(os.path.join(csrc, "Makefile")).read()
    flags = [ln for ln in mk.splitlines() if ln.startswith(("HIPFLAGS", "EMUFLAGS")) or "HIPFLAGS +=" in ln]
    assert len(flags) >= 3 and not any("-DNH" in ln for ln in flags), flags   # (-DNERFHIP_EMU: the emulator's; dbg / variant rules add theirs per command)
    assert "-DNH_DIAG $(DEFS)" in mk and "libnerfhip_$(NAME).so" in mk   # variants: marked, and never under the product's name
    assert "version() >= DIAG_FLAG and not ALLOW_DIAG" in open(os.path.join(pkg, "_lib.py")).read()
    wg = open(os.path.join(csrc, "wgrad.hip")).read()
    assert wg.count("#ifdef NH_WGRAD_TIMELINE") >= 3 and "nh_wall_clock()" in wg  # instrumentation is debug-build only
    assert sorted(f for f in os.listdir(csrc) if f.endswith(".hip")) == [
        "compact.hip", "dataio.hip", "elementwise.hip", "fused.hip", "mlp.hip", "mlp16.hip", "mlp16_ext.hip", "mlp16_w512.hip", "mlp64r.hip", "mlp_f16w.hip",
        "pack_f16.hip", "render.hip", "sample.hip", "wgrad.hip", "wgrad_f16.hip"]


def test_backward_mode_thresholds_and_option_plumbing():
    """TrainEngine(backward="auto") (round 6): the mode a net's next step runs in, from the zero-cotangent fraction its last compacted
    step reported -- dense while nothing is known or too little is dropped to pay for the gather; fp32 plans never the recomputing
    mode (a second fp32 forward over the kept samples costs more than the stash stream it saves); fp16-piece plans recompute from
    72 % dropped rows.  And the C-ABI option round-trips through the plan (no GPU needed: plans are host objects)."""
    from nerf_pytorch_amd.engine import TrainEngine as E
    assert [E._mode_for(f, False) for f in (None, 0.0, 0.10, 0.15, 0.5, 0.99)] == [0, 0, 0, 1, 1, 1]
    assert [E._mode_for(f, True) for f in (None, 0.0, 0.04, 0.05, 0.5, 0.71, 0.72, 0.99)] == [0, 0, 0, 1, 1, 1, 2, 2]
    lib = L.bind(L.LIB_PATH)
    cfg = L.ModelCfg(4, 128, 4, 10, 4, 1, 1, 1, 1, 1)
    plan = lib.plan_create(C.byref(cfg))
    assert plan and lib.plan_bwd_compaction(plan) == 0
    for mode in (1, 2, 0):
        lib.plan_set_bwd_compaction(plan, mode)
        assert lib.plan_bwd_compaction(plan) == mode
    with pytest.raises(L.NerfHipError, match="0 \\(dense\\), 1 \\(compacted\\), 2"):
        lib.plan_set_bwd_compaction(plan, 6)
    # the fused one-kernel backward (3, 4, 5; csrc/mlp64r.hip) exists for plans with an LDS-resident image only: a 4 x 128 net has none
    for mode in (3, 5):
        with pytest.raises(L.NerfHipError, match="fused backward"):
            lib.plan_set_bwd_compaction(plan, mode)
    fern = L.ModelCfg(4, 64, 3, 6, 4, 1, 1, 1, 1, 1)   # config/fern.yml's nets
    pf = lib.plan_create(C.byref(fern))
    for mode in (3, 4, 5, 0):
        lib.plan_set_bwd_compaction(pf, mode)
        assert lib.plan_bwd_compaction(pf) == mode
    # (mode 5's register-image stash -- 64 L + 192 floats per sample point -- lives in the plan's own stash region)
    assert lib.plan_stash_bytes(pf, 4096) >= 4096 * 4 * (64 * 4 + 192)
    # (its packed buffer carries the resident image behind the layer images; its scratch one partial per workgroup behind the list)
    assert lib.plan_packed_floats(pf) > 0 and lib.plan_bwd_scratch_bytes(pf, 4096) > lib.plan_bwd_stats_offset(pf, 4096)
    lib.plan_destroy(pf)
    # ... and TrainEngine(backward="auto") always runs it where it exists: over every sample (from the register-image stash where the
    # plan has it) until the list is known to drop 30 % (5 % against the recomputing mode 3)
    assert [E._mode_for(f, False, 3) for f in (None, 0.0, 0.04, 0.05, 0.9)] == [3, 3, 3, 4, 4]
    assert [E._mode_for(f, False, 5) for f in (None, 0.0, 0.05, 0.29, 0.30, 0.9)] == [5, 5, 5, 5, 4, 4]
    # the statistics words sit inside the backward scratch, behind everything the dense backward uses
    off, total = lib.plan_bwd_stats_offset(plan, 4096), lib.plan_bwd_scratch_bytes(plan, 4096)
    assert 0 < off < total and off % 4 == 0 and total - off >= 4 * (16 + 4096)
    lib.plan_destroy(plan)


def test_tolerance_table_holds_every_arithmetic_to_the_same_bounds():
    """the fp16-piece plans are held to the fp32 kernels' bounds -- the table has no per-arithmetic entry for an
    fp32-grade arithmetic, and the full-batch GPU tests select no bound (and no assertion) by arithmetic."""
    import re
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import tolerances as T
    for arith in T.FP32_GRADE:
        assert not T.OVERRIDES.get(arith), arith
        for name in T.TOL:
            assert T.bound(name, arith) == T.bound(name, "fp32"), (name, arith)
            assert isinstance(T.provenance(name), str) and len(T.provenance(name)) > 10, name
    src = open(os.path.join(ROOT, "tests", "test_gpu_fullsize.py")).read()
    # the only places the arithmetic's NAME may be compared in the suite's logic: the tags of the records
    for ln in src.splitlines():
        if re.search(r"\b(arith|infer)\s*[!=]=", ln):
            assert "self.tag" in ln or "self.name" in ln, ln
    # ... and no literal tolerance is left in its assertions: every bound comes out of the table
    body = src[src.index("def _end_to_end_on"):]
    lits = [ln.strip() for ln in body.splitlines() if ln.lstrip().startswith(("assert", "P.close")) and re.search(r"\d(\.\d+)?e-\d", ln.split("#")[0])]
    assert not lits, lits


def test_shard_bounds():
    from nerf_pytorch_amd.parallel import shard_bounds
    for n, w in ((8192, 8), (4096, 3), (10, 4), (3, 8)):
        prev = 0
        for r in range(w):
            lo, hi = shard_bounds(n, r, w)
            assert lo == prev and hi >= lo
            prev = hi
        assert prev == n


_DP_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "oracle"))
import nerf_oracle as O
from nerf_pytorch_amd.parallel import allreduce_gradients, shard_bounds
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
cfg = dict(num_layers=2, hidden_size=128, skip_connect_every=4, num_encoding_fn_xyz=4, num_encoding_fn_dir=2)
n, nc, nf = 8, 8, 8
g = torch.Generator().manual_seed(0)
ro = torch.tensor([0., 0., 4.]).expand(n, 3); rd = torch.randn(n, 3, generator=g) * 0.3; rd[:, 2] = -1
rays = O.pack_rays(ro, rd, 2.0, 6.0, rd); tgt = torch.rand(n, 3, generator=g)
rand = dict(t_rand=torch.rand(n, nc, generator=g), noise_coarse=torch.randn(n, nc, generator=g),
            u=torch.rand(n, nf, generator=g), noise_fine=torch.randn(n, nc + nf, generator=g))
opt = dict(num_coarse=nc, num_fine=nf, perturb=True, lindisp=False, white_background=False, noise_std=0.3)
def flat_grad(lo, hi):
    pc = {{k: v.requires_grad_(True) for k, v in O.init_params(cfg, 1).items()}}
    pf = {{k: v.requires_grad_(True) for k, v in O.init_params(cfg, 2).items()}}
    out = O.render_rays(rays[lo:hi], pc, pf, cfg, cfg, opt, {{k: v[lo:hi] for k, v in rand.items()}})
    loss, _, _, _ = O.loss_and_psnr(out["rgb_coarse"], out["rgb_fine"], tgt[lo:hi])
    loss.backward()
    return torch.cat([p.grad.reshape(-1) for p in list(pc.values()) + list(pf.values())])
lo, hi = shard_bounds(n, rank, world)
mine = flat_grad(lo, hi)
w = allreduce_gradients(mine)
mine /= w
full = flat_grad(0, n)
err = float((mine - full).abs().max() / full.abs().max())
assert w == world and err < 1e-5, err
# eval sharding (BASELINE config 5): ranks render disjoint, possibly ragged, row blocks; gather restores the image
from nerf_pytorch_amd.parallel import broadcast_parameters, gather_image_rows
H, W = 7, 5
img = torch.arange(H * W * 3, dtype=torch.float32).reshape(H, W, 3)
lo, hi = shard_bounds(H, rank, world)          # 4 + 3 rows
got = gather_image_rows(img[lo:hi].contiguous())
assert torch.equal(got, img), got.shape
lo, hi = shard_bounds(8, rank, world)          # equal shards take the single all_gather path
img8 = torch.arange(8 * W * 3, dtype=torch.float32).reshape(8, W, 3)
assert torch.equal(gather_image_rows(img8[lo:hi].contiguous()), img8)
flat = torch.full((11,), float(rank + 1))
broadcast_parameters(flat, src=0)
assert float(flat.sum()) == 11.0
dist.destroy_process_group()
print("rank", rank, "ok", err)
"""


def test_data_parallel_gradient_contract_gloo_world2(tmp_path):
    """Two gloo ranks each differentiate their ray shard (oracle), all-reduce the flat gradient and scale by 1/G: the
    result equals the single-process gradient over all rays (equal shards)."""
    script = tmp_path / "dp_worker.py"
    script.write_text(_DP_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


def test_bench_launches_its_own_ranks_when_started_plainly():
    """VERDICT r2 missing #1: `python bench.py --gpus 2` started directly (no torch.distributed.run around it) must become
    the launcher itself.  --dry-run keeps everything but the kernels: rendezvous, barrier, the per-rank timings gathered on
    rank 0, one JSON line -- on the CPU with gloo."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--global-rays", "8191"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["dry_run"] is True and j["value"] is None and len(j["ms_per_step_per_rank"]) == 2
    assert j["scaling"] == "strong"
    # the driver's own launch line (WORLD_SIZE set by torch.distributed.run) must not re-launch; a mismatch fails loudly
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"],
                         env=dict(env, WORLD_SIZE="1", RANK="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert bad.returncode != 0 and b"WORLD_SIZE=1" in bad.stderr
