"""One-process-per-GPU plumbing for the sharded inference path.

The path has no data-path collective (images are independent: per-image NMS, eval-mode BN,
SURVEY 8e); ``torch.distributed`` (backend 'nccl' = RCCL on ROCm, 'gloo' in CPU tests) is
only used to line ranks up around the timed region and to reduce the timing.  Mirrors what
the reference gets from ``mmcv.runner.init_dist(launcher, backend='nccl')``
(tools/test.py:156-160) + ``DistributedSampler(shuffle=False)`` (datasets/builder.py:112-113).
"""
import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the launcher's environment."""
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')),
            int(os.environ.get('WORLD_SIZE', '1')))


def init(backend='nccl', device=None):
    """Join the process group when launched with WORLD_SIZE > 1; returns (rank, local_rank, world)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        kwargs = {}
        if backend == 'nccl' and device is not None:
            kwargs['device_id'] = device
        dist.init_process_group(backend, **kwargs)
    return rank, local_rank, world


def shard(num_items, rank, world):
    """Contiguous, balanced [lo, hi) slice of ``num_items`` for ``rank`` (no padding, no overlap)."""
    base, extra = divmod(num_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def barrier(sync_device=True):
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if sync_device and torch.cuda.is_available():
        torch.cuda.synchronize()


def max_over_ranks(value, device='cpu'):
    """MAX all-reduce of a python float (the timed region's elapsed seconds)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def finalize():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
