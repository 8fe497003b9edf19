"""GPU-box diagnostic: k_mlp_fwd16 training variant (writes the stash) vs inference variant on the same input, back to
back, fine-pass size.  python scripts/fwd_train_vs_infer.py [variant-lib-name]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import nerf_pytorch_amd as N
if len(sys.argv) > 1 and sys.argv[1]:
    N._lib.LIB_PATH = os.path.join(ROOT, "nerf-pytorch_amd", "libnerfhip_%s.so" % sys.argv[1])
dev = torch.device("cuda", 0)
lib = N._lib.get_lib()
M = 786432
out = {}
for hid, lay in ((256, 8), (128, 4)):
    m = N.FlexibleNeRFModel(lay, hid, 4, 10, 4).to(dev)
    x = torch.randn(M, 90, device=dev)
    y = torch.empty(M, 4, device=dev)
    stash = torch.empty(int(lib.plan_stash_bytes(m._plan, M) // 4 * 1.1) + (1 << 22), device=dev)  # (slack: layout-experiment builds)
    st = torch.cuda.current_stream().cuda_stream
    packed = m._packed()
    macs = sum(p.numel() for n, p in m.named_parameters() if n.endswith("weight"))
    for name, sp in (("train", stash.data_ptr()), ("infer", None), ("train2", stash.data_ptr()), ("infer2", None)):
        for _ in range(3):
            lib.mlp_fwd(m._plan, packed.data_ptr(), x.data_ptr(), M, y.data_ptr(), sp, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            lib.mlp_fwd(m._plan, packed.data_ptr(), x.data_ptr(), M, y.data_ptr(), sp, st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        out["%dx%d %s" % (lay, hid, name)] = dict(ms=round(ms, 4), frac=round(2.0 * macs * M / (ms * 1e-3) / 157.3e12, 4))
print(json.dumps(out, indent=1))
