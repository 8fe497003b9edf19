"""The RCCL branch of TrainEngine on ONE GPU (tests/test_gpu_engine.py): a one-rank `nccl` process group with
`always_reduce=True`, so that every step issues the two asynchronous all-reduces (fine net's gradient before the coarse
backward, coarse net's after it) through RCCL's work handles and stream ordering -- what `bench.py --gpus N` runs with
N ranks.  A sum over one rank is the identity: parameters, Adam state and losses must equal, bit for bit, those of an
engine that has no process group at all; a wrong stream order (Adam before the collective, or the collective before the
weight-gradient reduction) would show up as a different result or as a race under the two-stream step.
Usage: python nccl_worker.py OVERLAP STEPS RAYS"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import nerf_oracle as O  # noqa: E402  (test infrastructure: deterministic initial weights)
import nerf_pytorch_amd as N  # noqa: E402

CFG = dict(num_layers=4, hidden_size=128, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)


def run(dev, overlap, steps, n, reduce):
    mc, mf = N.FlexibleNeRFModel(**CFG), N.FlexibleNeRFModel(**CFG)
    mc.load_state_dict(O.init_params(CFG, seed=1))
    mf.load_state_dict(O.init_params(CFG, seed=2))
    mc, mf = mc.to(dev), mf.to(dev)
    eng = N.TrainEngine(mc, mf, 32, 32, perturb=True, noise_std=0.2, lr=5e-3, seed=9, world_size=1, rank=0, overlap=overlap,
                        always_reduce=reduce)
    assert eng._reduce == reduce
    g = torch.Generator().manual_seed(3)
    ro = torch.tensor([0.0, 0.0, 4.0]).expand(n, 3)
    rd = torch.randn(n, 3, generator=g) * 0.3
    rd[:, 2] = -1.0
    rays = O.pack_rays(ro, rd, 2.0, 6.0, rd).to(dev)
    tgt = torch.rand(n, 3, generator=g).to(dev)
    losses = [eng.step(rays, tgt).clone() for _ in range(steps)]
    torch.cuda.synchronize()
    return mc.flat_params.clone(), mf.flat_params.clone(), eng.exp_avg.clone(), eng.exp_avg_sq.clone(), torch.stack(losses)


def main():
    overlap, steps, n = bool(int(sys.argv[1])), int(sys.argv[2]), int(sys.argv[3])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    plain = run(dev, overlap, steps, n, reduce=False)   # before any process group exists
    torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    assert torch.distributed.get_backend() == "nccl"
    reduced = run(dev, overlap, steps, n, reduce=True)
    for a, b, what in zip(plain, reduced, ("coarse params", "fine params", "exp_avg", "exp_avg_sq", "losses")):
        assert torch.equal(a, b), what + " differ between the RCCL path and the plain path"
    assert float(plain[4][-1, 2]) < float(plain[4][0, 2])
    torch.distributed.destroy_process_group()
    print("nccl one-rank engine ok: overlap=%d, %d steps, loss %.5f -> %.5f" % (overlap, steps, float(plain[4][0, 2]), float(plain[4][-1, 2])))


if __name__ == "__main__":
    main()
