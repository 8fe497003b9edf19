"""Data parallelism for the NeRF step: one process per GPU, replicated weights, rays sharded.

The path shards embarrassingly: every rank renders its own contiguous slice of the step's rays; the only exchange is
the all-reduce (sum) of the flat fp32 gradient (2 x 595,844 floats = 4.77 MB for 8x256) over RCCL/xGMI, issued as one
collective per net so that the fine net's half travels while the coarse net's backward still computes (engine.py).  Each rank's loss is a mean over its own rays, so for equal shards the global gradient is the rank
average: the 1/G factor is folded into the Adam kernel (engine.py).  Inference (eval) needs no collective at all.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_global, rank, world):
    """Contiguous ray shard [lo, hi) of `rank`; shards differ by at most one ray."""
    base, rem = divmod(n_global, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_gradients(flat_grad, group=None, async_op=False, single_rank=False):
    """In-place sum of a flat gradient (slice) across ranks (backend nccl == RCCL on ROCm; gloo in the CPU tests).
    The caller scales by 1/world (equal shards) when applying the update.
    async_op=False: returns the world size after the collective has been enqueued / completed.
    async_op=True: returns the work handle (None for one rank); `handle.wait()` orders the CURRENT stream behind the
    collective, so kernels launched in between (the coarse net's backward) overlap with it."""
    if not dist.is_available() or not dist.is_initialized():
        return None if async_op else 1
    world = dist.get_world_size(group)
    if world > 1 or single_rank:  # (single_rank: issue the collective even in a one-rank group -- the backend's path is
        #                             the same, the sum is the identity; TrainEngine(always_reduce=True))
        work = dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if async_op:
            return work
    return None if async_op else world


def broadcast_parameters(flat_params, src=0, group=None):
    """Make every rank start from rank `src`'s weights (only needed when ranks were not seeded identically)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_params, src=src, group=group)


def gather_image_rows(local_rows, group=None):
    """Eval (BASELINE config 5): ranks render disjoint row blocks of an image; rank 0 receives them in rank order.
    Pure output plumbing -- the render itself uses no collective."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local_rows
    world = dist.get_world_size(group)
    sizes = [torch.zeros(1, dtype=torch.int64, device=local_rows.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([local_rows.shape[0]], dtype=torch.int64, device=local_rows.device), group=group)
    if len({int(s) for s in sizes}) == 1:
        out = [torch.empty_like(local_rows, memory_format=torch.contiguous_format) for _ in sizes]
        dist.all_gather(out, local_rows.contiguous(), group=group)
    else:
        # ragged shards: pad to the largest, gather, trim
        mx = max(int(s) for s in sizes)
        pad = torch.zeros((mx, *local_rows.shape[1:]), dtype=local_rows.dtype, device=local_rows.device)
        pad[:local_rows.shape[0]] = local_rows
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad, group=group)
        out = [b[:int(s)] for b, s in zip(bufs, sizes)]
    return torch.cat(out, dim=0)
