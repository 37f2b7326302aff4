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


def backend_name():
    """The transport actually in use: 'nccl' (= RCCL on ROCm), 'gloo', ... or None without a process group."""
    if dist.is_available() and dist.is_initialized():
        return str(dist.get_backend())
    return None


def exchange_name(mode=None, world=None):
    """Human-readable description of the gradient exchange for benchmark records, derived from the live group."""
    world = world if world is not None else (dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1)
    if world <= 1:
        return 'one rank, no collective'
    mode = mode or os.environ.get('YV4_GRAD_EXCHANGE', 'allreduce')
    be = backend_name()
    lib = {'nccl': 'RCCL', 'gloo': 'gloo'}.get(be, str(be))
    how = {'allreduce': 'bucketed all-reduce (fp32)', 'direct': 'all-to-all reduce-scatter + all-gather (fp32)',
           'direct_bf16': 'all-to-all reduce-scatter + all-gather (bf16 wire, fp32 accumulate)'}[mode]
    return f'{lib} {how} of the flat gradient arena'


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
    """The one exchange step of data-parallel training: SUM of the gradients over ranks,
    then divide by the world size -- what the reference gets from ``MMDistributedDataParallel``
    (``mmdet/apis/train.py:74-82``), which copies ~430 gradient tensors into 25 MB buckets and back.

    Here the gradients already ARE one contiguous arena (``FlatState.grads``), so a bucket is a
    slice of it: no copy-in/copy-out, and few large collectives (xGMI is point-to-point, a ring
    all-reduce is per-link bound, so larger messages amortise the per-step latency better; 32 MB
    default -> 7 collectives for YOLOv4-L's 212 MB: the bucket that holds the first-registered parameters is final only
    when the stem's gradient is, so its exchange is the exposed tail -- measured with ``enable_timing`` at 64 MB: 0 % of
    backward left when it launches, 96 / 72 / 64 % for the three before it; halving the bucket halves that tail).  Buckets are launched from post-accumulate-grad
    hooks as soon as every gradient inside is final, i.e. overlapped with the rest of backward: arena
    order is registration order, backward produces the tail first, so the last bucket goes out first.

    ``mode``:
      'allreduce'    one ``all_reduce`` (RCCL's own algorithm choice) per bucket, fp32 on the wire.
      'direct'       reduce-scatter as ONE all-to-all (every rank sends chunk j of the bucket to rank j --
                     on a fully connected xGMI node all 7 links of a GPU carry payload at once instead of
                     the 2 a ring uses), the W received chunks are summed locally in fp32 and scaled by
                     1/W, then one all-gather returns the reduced chunks.  Bus bytes as the ring's
                     (2 (W-1)/W S), in 2 steps instead of 2 (W-1).
      'direct_bf16'  the same with bf16 on the wire (half the bytes; 105.8 MB instead of 211.7 MB for
                     YOLOv4-L): every rank rounds its own gradients to bf16 once, the sum over ranks is
                     accumulated in fp32, the reduced chunk is rounded to bf16 once more for the all-gather.
                     Relative error per element <= 2^-8 (two roundings), independent of the world size.
    The launch ORDER is the same on every rank by construction (strictly descending bucket index, a
    bucket waits for every bucket above it), so a rank with an unused parameter cannot enqueue
    collectives in another order than its peers.  Each parameter counts once per armed window; a second
    gradient event for a parameter whose bucket has already been exchanged (shared weights, re-entrant
    backward) raises instead of exchanging half a gradient -- use ``overlap=False`` for such models
    (everything is exchanged in ``finish()``).

    ``arm()`` before the backward whose gradients are to be exchanged (the last micro-batch of an
    accumulation window -- earlier micro-batches only accumulate locally), ``finish()`` after it.
    """

    MODES = ('allreduce', 'direct', 'direct_bf16')

    def __init__(self, flat, bucket_mb=32, group=None, mode=None, overlap=True, head_mb=8, exchange_at_world1=None):
        self.flat = flat
        self.group = group
        self.mode = mode or os.environ.get('YV4_GRAD_EXCHANGE', 'allreduce')
        if self.mode not in self.MODES:
            raise ValueError(f'GradReducer mode {self.mode!r} not in {self.MODES}')
        self.overlap = overlap
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        # a one-rank group still runs every collective (identity): the RCCL init, all_to_all_single /
        # all_gather_into_tensor and the side-stream ordering execute on a 1-GPU box (tests/test_gpu_ddp.py)
        if exchange_at_world1 is None:
            exchange_at_world1 = os.environ.get('YV4_EXCHANGE_AT_WORLD1') == '1'
        self.exchange = self.world > 1 or (bool(exchange_at_world1) and dist.is_available() and dist.is_initialized())
        self.buckets = self._make_buckets(flat, bucket_mb, head_mb)          # [lo, hi, [segment indices]]
        self._bucket_of = {}
        for bi, b in enumerate(self.buckets):
            for si in b[2]:
                self._bucket_of[si] = bi
        self._armed = False
        self._pending = []
        self._done = set()
        self._ready = []
        self._next = -1
        self._launched = []
        self._handles = []
        self._hooks = []
        self._index_of = {}
        self._stage = {}
        self._comm_stream = None
        self._timing = None          # enable_timing(): device events at arm / every bucket launch / finish
        for si, p in enumerate(flat._params):
            if p.requires_grad:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(si)))
                self._index_of[id(p)] = si
        # conv weights whose dW is accumulated straight into the arena bypass autograd's accumulation (and its
        # hooks): train_ops tells us when such a gradient is final
        from . import train_ops as _T
        # Weight gradients on train_ops' side stream TOGETHER with the exchange: built (the bucket's collective is ordered behind
        # the side stream in _launch) and measured on a one-rank RCCL group with every bucket's collective launched
        # (tools/ab_wstream_rccl.sh, profiles/r06_ab_wstream_rccl.txt): 1 175 images/s without the side stream, 1 122 with it
        # (all-reduce), 1 157 -> 1 080 (direct) -- a collective that has to wait for the side stream's queue starts later than
        # one behind the backward stream alone.  Off unless YV4_WGRAD_STREAM_WITH_EXCHANGE=1.
        self._direct_cb = _T.add_direct_grad_listener(
            self._on_direct_grad, side_stream_ok=os.environ.get('YV4_WGRAD_STREAM_WITH_EXCHANGE') == '1')

    @staticmethod
    def _make_buckets(flat, bucket_mb, head_mb):
        """Bucket boundaries by REVERSE cumulative size.  Backward finishes the arena from its end (arena order =
        registration order), so buckets are cut walking from the last segment towards the first, and the bucket of
        the FIRST-registered parameters -- final only when the stem's gradient is, i.e. the one exchange that cannot
        overlap backward -- is cut first and holds at most ``head_mb`` (8 MB for 32 MB buckets; it held whatever the
        forward walk left, 33 MB on YOLOv4-L, profiles/r03_train_overlap_bf16.json).  A single segment larger than a
        cap is a bucket by itself."""
        segs = flat.param_segments
        n = len(segs)
        ends = [segs[i + 1].offset if i + 1 < n else flat.n_param for i in range(n)]
        cap = max(4, int(bucket_mb * (1 << 20) // 4))
        head_cap = min(cap, max(4, int(head_mb * (1 << 20) // 4))) if head_mb else cap
        k = 0                                     # head bucket = segments [0, k)
        while k < n and (k == 0 or ends[k] - segs[0].offset <= head_cap):
            k += 1
        if k >= n:
            return [[segs[0].offset, ends[-1], list(range(n))]] if n else []
        rev, cur = [], None                       # the rest, cut from the end
        for si in range(n - 1, k - 1, -1):
            if cur is None or (cur[1] - segs[si].offset) > cap:
                cur = [segs[si].offset, ends[si], []]
                rev.append(cur)
            cur[0] = segs[si].offset
            cur[2].insert(0, si)
        return [[segs[0].offset, ends[k - 1], list(range(k))]] + rev[::-1]

    # ---- gradient-final events -------------------------------------------------------------------
    def _event(self, si):
        if not self._armed:
            return
        bi = self._bucket_of[si]
        if si in self._done:
            if self._launched[bi]:
                raise RuntimeError(
                    f'GradReducer: parameter {self.flat.param_segments[si].name} received a gradient after its bucket was '
                    'exchanged (shared weights or a re-entrant backward); construct the reducer with overlap=False')
            return
        self._done.add(si)
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._ready[bi] = True
            if self.overlap:
                self._cascade()

    def _on_direct_grad(self, p):
        si = self._index_of.get(id(p))
        if si is not None:
            self._event(si)

    def _make_hook(self, si):
        def hook(_p):
            self._event(si)
        return hook

    # ---- where in backward a bucket becomes exchangeable (measurement; works with one rank too) -------------------
    def enable_timing(self):
        """Record a device event when the window is armed, when each bucket is launched (its gradients are final: from
        here its exchange can run beside the rest of backward) and when ``finish()`` is called (end of backward)."""
        self._timing = dict(arm=None, finish=None, launch={})

    def timing_report(self):
        """After a synchronised step: per bucket (in launch order) its bytes and the fraction of the backward pass that
        was still ahead when it was launched = the share of backward its exchange can overlap; plus the byte-weighted
        mean.  With one rank nothing is exchanged, but the launch points are the same as on a node."""
        t = self._timing
        if not t or t['arm'] is None or t['finish'] is None:
            return None
        total = t['arm'].elapsed_time(t['finish'])
        rows, wsum, bsum, exposed = [], 0.0, 0, 0
        for bi in self.launch_order:
            lo, hi, _ = self.buckets[bi]
            ev = t['launch'].get(bi)
            ahead = max(0.0, ev.elapsed_time(t['finish'])) / total if ev is not None and total > 0 else 0.0
            rows.append(dict(bucket=bi, mbytes=round((hi - lo) * 4 / 1e6, 1), backward_ahead=round(ahead, 3),
                             ahead_ms=round(ahead * total, 2)))
            wsum += ahead * (hi - lo)
            bsum += hi - lo
            # a bucket's exchange is EXPOSED when backward ends before it can: at a (pessimistic) 100 GB/s of ring
            # bus bandwidth plus 0.1 ms of launch latency per collective
            need_ms = 0.1 + (hi - lo) * 4 * 2 / 100e9 * 1e3
            if ahead * total < need_ms:
                exposed += hi - lo
        return dict(backward_ms=round(total, 2), buckets=rows, overlappable_fraction=round(wsum / max(bsum, 1), 3),
                    exposed_mbytes=round(exposed * 4 / 1e6, 1),
                    exchanged_mbytes=round(bsum * 4 / 1e6, 1), mode=self.mode, world=self.world)

    def _stamp(self, what, bi=None):
        if self._timing is None or not self.flat.grads.is_cuda:
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        if bi is None:
            self._timing[what] = ev
        else:
            self._timing['launch'][bi] = ev

    def arm(self):
        self._armed = True
        if self._timing is not None:
            self._timing['launch'] = {}
            self._stamp('arm')
        self._pending = [sum(1 for si in b[2] if self.flat._params[si].requires_grad) for b in self.buckets]
        self._done = set()
        self._ready = [n == 0 for n in self._pending]
        self._launched = [False] * len(self.buckets)
        self._next = len(self.buckets) - 1
        self._handles = []
        self.launch_order = []

    def _cascade(self):
        """Launch every ready bucket whose successors have all been launched: strictly descending order."""
        while self._next >= 0 and self._ready[self._next]:
            self._launch(self._next)
            self._next -= 1

    # ---- the exchange ------------------------------------------------------------------------------
    def _staging(self, bi, n):
        st = self._stage.get(bi)
        if st is None:
            wire = torch.bfloat16 if self.mode == 'direct_bf16' else torch.float32
            chunk = -(-n // self.world)
            chunk += (-chunk) % 8                     # 16-byte aligned chunks in either wire type
            dev = self.flat.grads.device
            st = dict(chunk=chunk, send=torch.zeros(self.world * chunk, dtype=wire, device=dev),
                      recv=torch.empty(self.world * chunk, dtype=wire, device=dev),
                      mine=torch.empty(chunk, dtype=wire, device=dev),
                      out=torch.empty(self.world * chunk, dtype=wire, device=dev))
            self._stage[bi] = st
        return st

    def _direct_exchange(self, bi):
        lo, hi, _ = self.buckets[bi]
        n = hi - lo
        st = self._staging(bi, n)
        g = self.flat.grads[lo:hi]
        st['send'][:n].copy_(g)                        # (rounds to bf16 in 'direct_bf16'); the pad stays zero
        dist.all_to_all_single(st['recv'], st['send'], group=self.group)
        red = st['recv'].view(self.world, st['chunk']).sum(dim=0, dtype=torch.float32).mul_(1.0 / self.world)
        st['mine'].copy_(red)
        dist.all_gather_into_tensor(st['out'], st['mine'], group=self.group)
        g.copy_(st['out'][:n])

    def _launch(self, bi):
        if self._launched[bi]:
            return
        self._launched[bi] = True
        self.launch_order.append(bi)
        self._stamp('launch', bi)
        if not self.exchange:
            return
        lo, hi, _ = self.buckets[bi]
        # Weight gradients of this bucket may have been launched on train_ops' side stream (they are, by default): the bucket's
        # exchange is ordered behind BOTH the backward stream and that side stream, on a stream of its own -- the backward
        # stream itself waits for nothing here (round 5 switched the side stream off whenever a reducer existed, i.e. on every
        # multi-GPU run).
        from . import train_ops as _T
        cuda = self.flat.grads.is_cuda
        side = _T.wgrad_side_stream(self.flat.grads.device) if cuda else None
        if self.mode == 'allreduce':
            if side is not None:
                if self._comm_stream is None:
                    self._comm_stream = torch.cuda.Stream(device=self.flat.grads.device)
                self._comm_stream.wait_stream(torch.cuda.current_stream())
                self._comm_stream.wait_stream(side)
                with torch.cuda.stream(self._comm_stream):      # (the collective's own stream waits for the stream current HERE)
                    self._handles.append(dist.all_reduce(self.flat.grads[lo:hi], op=dist.ReduceOp.SUM, group=self.group,
                                                         async_op=True))
                return
            self._handles.append(dist.all_reduce(self.flat.grads[lo:hi], op=dist.ReduceOp.SUM, group=self.group,
                                                 async_op=True))
            return
        if cuda:
            # the staging copies and the local fp32 sum run on a side stream behind the producing kernels, so the
            # backward stream never waits for a collective before finish()
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=self.flat.grads.device)
            self._comm_stream.wait_stream(torch.cuda.current_stream())
            if side is not None:
                self._comm_stream.wait_stream(side)
            with torch.cuda.stream(self._comm_stream):
                self._direct_exchange(bi)
        else:
            self._direct_exchange(bi)

    def finish(self):
        """Exchange whatever backward has not triggered (unused parameters), wait, average."""
        if not self._armed:
            raise RuntimeError('GradReducer.finish() without arm()')
        self._stamp('finish')
        self._ready = [True] * len(self.buckets)
        self._cascade()
        for h in self._handles:
            h.wait()
        self._handles = []
        self._armed = False
        if self.exchange:
            if self.mode == 'allreduce':
                self.flat.grads.mul_(1.0 / self.world)
            elif self._comm_stream is not None:
                torch.cuda.current_stream().wait_stream(self._comm_stream)

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        from . import train_ops as _T
        _T.remove_direct_grad_listener(self._direct_cb)


def finalize():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
