"""Diagnostic: per-tensor gradient errors of case_mlp_backward-like runs for a precision, geometry, g_scale, w_gain."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import nerf_oracle as O, parity_cases as P, backends
b = backends.EmuBackend() if os.environ.get("NH_DEBUG_EMU") else backends.GpuBackend()
name, m = sys.argv[1], int(sys.argv[2])
for prec, g_scale, w_gain in [(int(x.split(":")[0]), float(x.split(":")[1]), float(x.split(":")[2])) for x in sys.argv[3:]]:
    cfg = P.MLP_GEOMETRIES[name]
    plan, params, flat, packed = P.mlp_setup(b, cfg, seed=41, precision=prec, w_gain=w_gain)
    dx, dd = O.model_dims(cfg)
    gen = P.rng(42)
    x = torch.randn(m, dx + dd, generator=gen)
    go = torch.randn(m, 4, generator=gen) * g_scale
    keep = O.mlp_relu_margin(params, x, cfg) > 1e-6
    x, go = x[keep].contiguous(), go[keep].contiguous()
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    (O.mlp_forward(p, x, cfg) * go).sum().backward()
    _, stash = b.mlp_fwd(plan, packed, x.numpy(), want_stash=True)
    grads = b.unflatten(plan, b.mlp_bwd(plan, packed, go.numpy(), stash))
    print("precision %d g_scale %g w_gain %g rows %d" % (prec, g_scale, w_gain, x.shape[0]))
    for k, v in p.items():
        ref = v.grad.numpy(); sc = float(np.abs(ref).max()) + 1e-30
        print("   %-22s max|g| %.3e  err/max|g| %.3e  finite %s" % (k, sc, float(np.abs(grads[k] - ref).max()) / sc, bool(np.isfinite(grads[k]).all())))
    b.lib.plan_destroy(plan)
