"""GPU-box measurement: the inference forward of one network over M sample points, fp32 plan (k_mlp_fwd16<W, VIEW, false>)
beside the NERFHIP_PRECISION_BF16X3 plan (k_mlp_fwd_bf16x3<W, VIEW>), through the C ABI (nerfhip_mlp_fwd, caller-encoded
rows), timed with events on the launch stream.  Prints fp32-equivalent TFLOP/s (2 x weights per sample point) and each
variant's fraction of ITS OWN roofline (157.3 TF fp32 MFMA; 2500 / 3 TF for three bf16 MFMAs per product block)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import nerf_pytorch_amd as N  # noqa: E402
import nerf_pytorch_amd._lib as L  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    if len(sys.argv) > 2:  # an A/B build of the library (scripts/build_bf16_variant.sh)
        L.LIB_PATH = os.path.join(ROOT, "nerf-pytorch_amd", sys.argv[2])
        print("# " + sys.argv[2])
    lib = L.get_lib()
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096 * 192
    reps = 10
    for layers, hidden in ((8, 256), (4, 128)):
        mc = L.ModelCfg(layers, hidden, 4, 10, 4, 1, 1, 1, 1, 1)
        res = {}
        for prec, name in ((0, "fp32"), (1, "bf16x3")):
            torch.manual_seed(0)
            plan = lib.plan_create_ex(C.byref(mc), prec)
            assert plan, lib.last_error()
            nparams = lib.plan_num_params(plan)
            flat = (torch.randn(nparams, device=dev) * 0.05).contiguous()
            n = lib.plan_packed_floats(plan)
            table = torch.empty(n, dtype=torch.int32)
            lib.plan_pack_index(plan, table.data_ptr())
            table = table.to(dev)
            packed = torch.empty(n, dtype=torch.float32, device=dev)
            d = lib.plan_dim_xyz(plan) + lib.plan_dim_dir(plan)
            x = torch.randn(M, d, device=dev)
            out = torch.empty(M, 4, device=dev)
            weights = sum(r * c for (_, _, r, c) in
                          [(lambda nm, off, rows, cols: (nm.value, off.value, rows.value, cols.value))(*t) for t in
                           [_tensor(lib, plan, i) for i in range(lib.plan_num_tensors(plan))]] if c)
            with L.launch_on(flat, table, packed, x, out) as st:
                lib.pack_weights_plan(plan, flat.data_ptr(), table.data_ptr(), packed.data_ptr(), st)
                for _ in range(2):
                    lib.mlp_fwd(plan, packed.data_ptr(), x.data_ptr(), M, out.data_ptr(), None, st)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    lib.mlp_fwd(plan, packed.data_ptr(), x.data_ptr(), M, out.data_ptr(), None, st)
                e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            tf = 2.0 * weights * M / (ms * 1e-3) / 1e12
            peak = 157.3 if prec == 0 else 2500.0 / 3.0
            res[name] = (ms, tf, out.clone())
            print("%dx%d %-7s M=%d  %8.3f ms  %7.1f fp32-equivalent TFLOP/s  = %.3f of its own roofline (%.1f TF), %.2f x the fp32 "
                  "MFMA peak" % (layers, hidden, name, M, ms, tf, tf / peak, peak, tf / 157.3), flush=True)
            lib.plan_destroy(plan)
        a, b = res["fp32"][2], res["bf16x3"][2]
        print("    bf16x3 vs fp32 outputs (same random weights and rows): max |diff| / max |out| = %.3e; speed-up %.2fx" %
              (float((a - b).abs().max() / a.abs().max()), res["fp32"][0] / res["bf16x3"][0]), flush=True)


def _tensor(lib, plan, i):
    nm, off, rows, cols = C.c_char_p(), C.c_int64(), C.c_int(), C.c_int()
    lib.plan_tensor_info(plan, i, C.byref(nm), C.byref(off), C.byref(rows), C.byref(cols))
    return nm, off, rows, cols


if __name__ == "__main__":
    main()
