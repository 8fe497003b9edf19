"""One rank of the world-size-2 TrainEngine test (tests/test_gpu_engine.py): both ranks run on cuda:0 with a gloo process
group (RCCL refuses two ranks on one device; the engine code path -- sharded ray selection, per-net asynchronous gradient
all-reduce overlapped with the coarse backward, 1/G folded into Adam -- is the one bench.py --gpus N runs).
Usage: RANK=r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=p python dp_worker.py OUT_DIR STEPS RAYS_PER_RANK OVERLAP"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import nerf_oracle as O  # noqa: E402  (test infrastructure: deterministic initial weights)
import nerf_pytorch_amd as N  # noqa: E402

CFG = dict(num_layers=4, hidden_size=128, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)
H, W, FOCAL, NC, NF = 40, 40, 55.0, 32, 32


def scene(dev):
    g = torch.Generator().manual_seed(77)
    image = torch.rand(H, W, 3, generator=g).to(dev)
    pose = torch.eye(4)
    pose[2, 3] = 4.0
    return image, pose.to(dev)


def make_engine(dev, world, rank, overlap, pg=None):
    mc, mf = N.FlexibleNeRFModel(**CFG), N.FlexibleNeRFModel(**CFG)
    mc.load_state_dict(O.init_params(CFG, seed=1))
    mf.load_state_dict(O.init_params(CFG, seed=2))
    mc, mf = mc.to(dev), mf.to(dev)
    eng = N.TrainEngine(mc, mf, NC, NF, perturb=True, noise_std=0.2, lr=5e-3, seed=9, world_size=world, rank=rank,
                        overlap=overlap, process_group=pg)
    return mc, mf, eng


def run(eng, image, pose, steps, n, dev):
    """First step in two halves so that the all-reduced gradient can be recorded, then `steps - 1` whole steps."""
    from nerf_pytorch_amd.train_utils import select_training_rays
    opts = N.make_options(NC, NF)
    rays, tgt, _ = select_training_rays(H, W, FOCAL, pose, image, n, opts, seed=eng.seed, step=eng.step_count,
                                        first=eng.rank * n)
    eng.forward_backward(rays, tgt, ray_offset=eng.rank * n)
    eng.wait_gradients()
    torch.cuda.synchronize()
    grad0 = (eng.grad / eng.world).cpu().numpy()
    eng.optimizer_step()
    for _ in range(steps - 1):
        eng.step_on_image(image, pose, H, W, FOCAL, opts, n)
    torch.cuda.synchronize()
    return grad0, eng.mc.flat_params.cpu().numpy(), eng.mf.flat_params.cpu().numpy(), eng.loss.cpu().numpy()


def main():
    out_dir, steps, n, overlap = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), bool(int(sys.argv[4]))
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    image, pose = scene(dev)
    _, _, eng = make_engine(dev, world, rank, overlap)
    grad0, pc, pf, loss = run(eng, image, pose, steps, n, dev)
    # the self-diagnosis of bench.py's N-GPU line (TrainEngine.collective_times_ms): times the exchange on a zero scratch, after the
    # step's own collectives -- the live gradient must come back untouched and finite (it used to be summed world^23 times)
    before = eng.grad.clone()
    coll = eng.collective_times_ms(reps=2, warmup=1)
    assert torch.equal(eng.grad, before) and bool(torch.isfinite(eng.grad).all()), "collective_times_ms touched the gradient"
    assert coll["fine"] is not None and coll["coarse"] is not None and coll["fine"] >= 0.0 and coll["coarse"] >= 0.0, coll
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), grad0=grad0, pc=pc, pf=pf, loss=loss, coll=np.array([coll["fine"], coll["coarse"]]))
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
