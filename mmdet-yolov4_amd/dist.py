"""One-process-per-GPU plumbing: the sharded inference path and the gradient exchange of the
training step.

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
    """Join the process group when launched with WORLD_SIZE > 1; returns (rank, local_rank, world).
    YV4_DIST_FORCE_INIT=1 joins a one-rank group too (exercises the RCCL barrier / all-reduce of the timed
    region on a 1-GPU box); YV4_DIST_BACKEND overrides the backend (tests: gloo with both ranks on cuda:0)."""
    rank, local_rank, world = env_world()
    backend = os.environ.get('YV4_DIST_BACKEND', backend)     # e.g. gloo: two ranks sharing the one GPU of a test box
    force = os.environ.get('YV4_DIST_FORCE_INIT') == '1' and 'MASTER_PORT' in os.environ
    if (world > 1 or force) and not dist.is_initialized():
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


def sampler_indices(num_items, rank, world):
    """The dataset indices rank ``rank`` evaluates in a distributed test run: the order of
    ``DistributedSampler(shuffle=False)`` (datasets/builder.py:112-113) -- the index list is padded with its own
    head to a multiple of ``world`` and dealt round-robin -- which is what ``collect_results`` undoes."""
    per_rank = -(-num_items // world)
    padded = [i % max(num_items, 1) for i in range(per_rank * world)] if num_items else []
    return padded[rank::world]


def collect_results(result_part, size, device='cpu'):
    """Merge every rank's list of per-image results on rank 0, in dataset order (other ranks get None).

    The result-gather step of a distributed test run (mmdet/apis/test.py:160-191, ``collect_results_gpu``):
    results are variable-length python objects, so each rank ships one byte string -- an all-gather of the
    lengths, then one all-gather of the parts padded to the longest -- and rank 0 re-interleaves them (rank r
    holds items r, r+world, ...) and drops the sampler's padding beyond ``size``.  Host-side; ``device`` is
    where the byte buffers live ('cpu' for gloo, the rank's GPU for RCCL)."""
    import pickle
    if not (dist.is_available() and dist.is_initialized()):
        return list(result_part)[:size]
    rank, world = dist.get_rank(), dist.get_world_size()
    part = torch.frombuffer(bytearray(pickle.dumps(result_part)), dtype=torch.uint8).to(device)
    length = torch.tensor([part.numel()], dtype=torch.int64, device=device)
    lengths = [torch.zeros_like(length) for _ in range(world)]
    dist.all_gather(lengths, length)
    longest = int(max(int(l) for l in lengths))
    send = torch.zeros(longest, dtype=torch.uint8, device=device)
    send[:part.numel()] = part
    recv = [torch.zeros_like(send) for _ in range(world)]
    dist.all_gather(recv, send)
    if rank != 0:
        return None
    parts = [pickle.loads(r[:int(l)].cpu().numpy().tobytes()) for r, l in zip(recv, lengths)]
    ordered = []
    for group in zip(*parts):
        ordered.extend(group)
    return ordered[:size]


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


class GradReducer:
    """The one exchange step of data-parallel training: SUM all-reduce of the
    gradients, then divide by the world size -- what the reference gets from
    ``MMDistributedDataParallel`` (``mmdet/apis/train.py:96-103``), which copies ~430 gradient
    tensors into 25 MB buckets and back.

    Here the gradients already ARE one contiguous arena (``FlatState.grads``), so a bucket is a
    slice of it: no copy-in/copy-out, and few large collectives (xGMI is point-to-point, a ring
    all-reduce is per-link bound, so larger messages amortise the per-step latency better; 64 MB
    default -> 4 collectives for YOLOv4-L's 212 MB).  Buckets are launched asynchronously from
    post-accumulate-grad hooks as soon as every gradient inside is final, i.e. overlapped with the
    rest of backward: arena order is registration order, backward produces the tail first, so
    the last bucket goes out first.

    ``arm()`` before the backward whose gradients are to be exchanged (the last micro-batch of an
    accumulation window -- earlier micro-batches only accumulate locally), ``finish()`` after it.
    """

    def __init__(self, flat, bucket_mb=64, group=None):
        self.flat = flat
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        cap = max(4, int(bucket_mb * (1 << 20) // 4))
        self.buckets = []          # [lo, hi, [segment indices]]
        cur = None
        for si, seg in enumerate(flat.param_segments):
            end = flat.param_segments[si + 1].offset if si + 1 < len(flat.param_segments) else flat.n_param
            if cur is None or (end - cur[0]) > cap and cur[2]:
                cur = [seg.offset, end, []]
                self.buckets.append(cur)
            cur[1] = end
            cur[2].append(si)
        self._bucket_of = {}
        for bi, b in enumerate(self.buckets):
            for si in b[2]:
                self._bucket_of[si] = bi
        self._armed = False
        self._pending = []
        self._launched = []
        self._handles = []
        self._hooks = []
        self._index_of = {}
        for si, p in enumerate(flat._params):
            if p.requires_grad:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(si)))
                self._index_of[id(p)] = si
        # conv weights whose dW is accumulated straight into the arena bypass autograd's accumulation (and its
        # hooks): train_ops tells us when such a gradient is final
        from . import train_ops as _T
        self._direct_cb = _T.add_direct_grad_listener(self._on_direct_grad)

    def _on_direct_grad(self, p):
        si = self._index_of.get(id(p))
        if si is not None and self._armed:
            bi = self._bucket_of[si]
            self._pending[bi] -= 1
            if self._pending[bi] == 0:
                self._launch(bi)

    def _make_hook(self, si):
        def hook(_p):
            if self._armed:
                bi = self._bucket_of[si]
                self._pending[bi] -= 1
                if self._pending[bi] == 0:
                    self._launch(bi)
        return hook

    def arm(self):
        self._armed = True
        self._pending = [sum(1 for si in b[2] if self.flat._params[si].requires_grad) for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._handles = []

    def _launch(self, bi):
        if self._launched[bi]:
            return
        self._launched[bi] = True
        if self.world > 1:
            lo, hi, _ = self.buckets[bi]
            self._handles.append(dist.all_reduce(self.flat.grads[lo:hi], op=dist.ReduceOp.SUM, group=self.group,
                                                 async_op=True))

    def finish(self):
        """Exchange whatever backward has not triggered (unused parameters), wait, average."""
        if not self._armed:
            raise RuntimeError('GradReducer.finish() without arm()')
        for bi in range(len(self.buckets)):
            self._launch(bi)
        for h in self._handles:
            h.wait()
        self._handles = []
        self._armed = False
        if self.world > 1:
            self.flat.grads.mul_(1.0 / self.world)

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        from . import train_ops as _T
        _T.remove_direct_grad_listener(self._direct_cb)


def finalize():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
