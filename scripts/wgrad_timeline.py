"""GPU-box diagnostic: per-workgroup timeline of k_wgrad for one fine-pass-sized launch (M = 786432 samples)."""
import ctypes as C, os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nerf_pytorch_amd as N
N._lib.LIB_PATH = os.path.join(ROOT, "nerf-pytorch_amd", "libnerfhip_dbg.so")  # make -C nerf-pytorch_amd/csrc dbg
dev = torch.device("cuda", 0)
lib = N._lib.get_lib()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 786432
HID = int(sys.argv[2]) if len(sys.argv) > 2 else 256
LAY = int(sys.argv[3]) if len(sys.argv) > 3 else 8
COSTS_OUT = sys.argv[4] if len(sys.argv) > 4 else None
m = N.FlexibleNeRFModel(LAY, HID, 4, 10, 4).to(dev)
x = torch.randn(M, 90, device=dev)
out = torch.empty(M, 4, device=dev)
stash = torch.empty(lib.plan_stash_bytes(m._plan, M) // 4, device=dev)
st = torch.cuda.current_stream().cuda_stream
packed = m._packed()
lib.mlp_fwd(m._plan, packed.data_ptr(), x.data_ptr(), M, out.data_ptr(), stash.data_ptr(), st)
g = torch.randn(M, 4, device=dev)
sb = lib.plan_bwd_scratch_bytes(m._plan, M)
scratch = torch.zeros(sb // 4 + 1, device=dev)
gp = torch.empty(m.num_flat_params, device=dev)
for _ in range(2):
    lib.mlp_bwd(m._plan, packed.data_ptr(), g.data_ptr(), M, stash.data_ptr(), scratch.data_ptr(), sb, gp.data_ptr(), st)
torch.cuda.synchronize()
nt = 4 * ((M + 127) // 128)
KW = 128 if HID <= 128 else 256                    # kernel width; a partial = the largest job's tiles + 512 bias + 128 records
rows = LAY * KW + KW + KW // 2 + 32
buf = C.create_string_buffer(1 << 14)
lib.plan_describe(m._plan, buf, len(buf))
tiles = 1
for line in buf.value.decode().splitlines()[1:]:
    f = dict(zip(line.split()[0::2], line.split()[1::2]))
    ta, tb = (int(v) for v in f["tiles"].split("x"))
    tiles = max(tiles, ta * tb + (tb if f["side"] == "1" else (ta * int(f["side_tiles"]) if f["side"] == "2" else 0)))
print(buf.value.decode())
BIAS = tiles * 1024                                # a partial = the largest job's accumulator tiles (its side tiles included)
PART = BIAS + 512 + 128
part = scratch[nt * rows * 32:]
nwg = part.numel() // PART
part = part[:nwg * PART].view(nwg, PART)[:, BIAS + 512:BIAS + 512 + 128].contiguous().cpu().numpy().view(np.uint64).reshape(nwg, 8, 8)
t0 = part[:, :, 0][part[:, :, 0] > 0].min()
jobs = {}
for w in range(nwg):
    act = part[w, :, 1] > 0
    if not act.any():
        continue
    b = (part[w, act, 0].min() - t0) / 100.0   # us (100 MHz)
    e = (part[w, act, 1].max() - t0) / 100.0
    per_wave = (part[w, act, 1] - part[w, act, 0]) / 100.0
    j = int(part[w, act, 2][0])
    jobs.setdefault(j, []).append((b, e, per_wave.mean(), int(act.sum())))
end = max(e for v in jobs.values() for (_, e, _, _) in v)
ok = part[:, :, 1] > part[:, :, 0]
ghz = ((part[:, :, 5] - part[:, :, 4])[ok] / ((part[:, :, 1] - part[:, :, 0])[ok] * 10.0))   # core cycles per ns
print("# instrumented build (make dbg): the per-piece descriptor re-uniformisation it needs (wgrad.hip, WStageDma::issue) makes "
      "this kernel ~10 % slower than the product's, small jobs more: read the ORDER of events and the wait / barrier shares, "
      "not absolute times")
print(json.dumps(dict(nwg=nwg, makespan_us=end, core_clock_ghz_mean=float(ghz.mean()), core_clock_ghz_min=float(ghz.min()),
                      core_clock_ghz_max=float(ghz.max()))))
for j in sorted(jobs):
    v = np.array(jobs[j])
    print("job %2d  wgs %4d  waves/wg %d  start %8.1f..%8.1f us  end %8.1f..%8.1f us  dur mean %8.1f max %8.1f us" % (
        j, len(v), int(v[0, 3]), v[:, 0].min(), v[:, 0].max(), v[:, 1].min(), v[:, 1].max(), (v[:, 1] - v[:, 0]).mean(), (v[:, 1] - v[:, 0]).max()))
# per-job time per sample tile of one workgroup -> the split-K costs that would equalise workgroup durations
# share of a wave's cycles parked in s_waitcnt vmcnt(0) / s_barrier at the stage boundaries, per job
for j in sorted(jobs):
    sel = (part[:, :, 1] > 0) & (part[:, :, 2] == j)
    tot = (part[:, :, 5] - part[:, :, 4])[sel].astype(np.float64)
    wait = (part[:, :, 6] >> np.uint64(32))[sel].astype(np.float64)
    bar = (part[:, :, 6] & np.uint64(0xffffffff))[sel].astype(np.float64)
    loop = part[:, :, 7][sel].astype(np.float64)
    print("job %2d  wave-cycle shares: vmcnt wait %.3f  barrier %.3f  k-step loop %.3f  rest %.3f" % (
        j, (wait / tot).mean(), (bar / tot).mean(), (loop / tot).mean(), 1 - ((wait + bar + loop) / tot).mean()))
tau = {}
for j in sorted(jobs):
    v = np.array(jobs[j])
    tau[j] = float((v[:, 1] - v[:, 0]).mean()) * len(v) / nt
scale = 1000.0 / max(tau.values())
costs = [max(1, int(round(tau[j] * scale))) for j in sorted(tau)]
print("us per sample tile per workgroup:", " ".join("%d:%.3f" % (j, tau[j]) for j in sorted(tau)))
print("fitted costs:", ",".join(str(c) for c in costs))
if COSTS_OUT:
    open(COSTS_OUT, "w").write(",".join(str(c) for c in costs))
