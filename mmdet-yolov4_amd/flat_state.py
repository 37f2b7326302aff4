"""Flat fp32 arenas for the training step.

The reference keeps every parameter, gradient, momentum buffer and EMA copy as its own
allocation and walks them in Python: one param group per parameter for SGD
(``core/custom_hooks/warmup_hooks.py:24-32``), a loop over all 658 state entries for the EMA
(``core/custom_hooks/ema_hooks.py:80-98``), a bucketed copy-in/copy-out for the DDP all-reduce.
On a 288 GB part there is no reason to: ``FlatState`` re-homes the model's state into

  values  [ parameters | float buffers (BN running stats) ]   one fp32 allocation
  grads   [ parameters ]                                       one fp32 allocation
  ints    [ integer buffers (num_batches_tracked) ]            one int64 allocation

Every ``nn.Parameter`` / buffer of the model becomes a *view* of its slice (names, shapes,
``state_dict`` keys and their order are unchanged), so the optimizer step, gradient clipping,
the EMA and the gradient all-reduce are each one streaming pass (``csrc/optim.hip``) or one
collective over a contiguous range.  Slices start at multiples of 4 floats (16-byte accesses).

Conv weights are stored ``channels_last`` (physically ``(Cout, KH, KW, Cin)``): that is the
packed K order ``(kh, kw, ci)`` of the conv kernels, so packing a weight for a launch and
receiving ``dW`` from ``yv4_conv_wgrad`` are views, not copies.
"""
import torch
import torch.nn as nn


def _pad4(n):
    return (n + 3) // 4 * 4


class Segment:
    __slots__ = ('name', 'offset', 'numel', 'shape', 'kind')

    def __init__(self, name, offset, numel, shape, kind):
        self.name, self.offset, self.numel, self.shape, self.kind = name, offset, numel, tuple(shape), kind

    def __repr__(self):
        return f'Segment({self.name}, off={self.offset}, n={self.numel}, {self.kind})'


def _view_as_param(flat, seg):
    """View of a flat slice with the logical shape of the tensor; 4-D tensors (conv weights)
    are laid out channels_last."""
    sl = flat[seg.offset:seg.offset + seg.numel]
    if len(seg.shape) == 4:
        o, i, kh, kw = seg.shape
        return sl.view(o, kh, kw, i).permute(0, 3, 1, 2)
    return sl.view(seg.shape)


class FlatState:
    """Re-home ``model``'s parameters and buffers into flat arenas (in place)."""

    def __init__(self, model, skip_buffer_prefix='ema_'):
        if getattr(model, '_flat_state', None) is not None:
            raise RuntimeError('the model already has a FlatState (use FlatState.of(model))')
        params = [(n, p) for n, p in model.named_parameters()]
        if not params:
            raise ValueError('FlatState: the model has no parameters')
        dev = params[0][1].device
        for n, p in params:
            if p.dtype != torch.float32:
                raise TypeError(f'FlatState: parameter {n} is {p.dtype}; the arenas are fp32')
            if p.device != dev:
                raise ValueError('FlatState: all parameters must live on one device')
        fbufs, ibufs = [], []
        for n, b in model.named_buffers():
            if n.split('.')[-1].startswith(skip_buffer_prefix) or b is None:
                continue
            (fbufs if b.dtype.is_floating_point else ibufs).append((n, b))

        self.device = dev
        self.param_segments, self.buffer_segments, self.int_segments = [], [], []
        off = 0
        for n, p in params:
            self.param_segments.append(Segment(n, off, p.numel(), p.shape, 'param'))
            off += _pad4(p.numel())
        self.n_param = off
        for n, b in fbufs:
            self.buffer_segments.append(Segment(n, off, b.numel(), b.shape, 'buffer'))
            off += _pad4(b.numel())
        self.n_state = off
        ioff = 0
        for n, b in ibufs:
            self.int_segments.append(Segment(n, ioff, b.numel(), b.shape, 'int'))
            ioff += b.numel()
        self.n_int = ioff

        self.values = torch.zeros(max(self.n_state, 4), dtype=torch.float32, device=dev)
        self.grads = torch.zeros(max(self.n_param, 4), dtype=torch.float32, device=dev)
        self.ints = torch.zeros(max(self.n_int, 1), dtype=torch.int64, device=dev)

        self._params = []
        with torch.no_grad():
            for seg, (n, p) in zip(self.param_segments, params):
                v = _view_as_param(self.values, seg)
                v.copy_(p.data)
                p.data = v
                self._params.append(p)
            mods = dict(model.named_modules())
            for seg, (n, b) in zip(self.buffer_segments, fbufs):
                v = _view_as_param(self.values, seg)
                v.copy_(b)
                self._set_buffer(mods, n, v)
            for seg, (n, b) in zip(self.int_segments, ibufs):
                v = self.ints[seg.offset:seg.offset + seg.numel].view(seg.shape)
                v.copy_(b)
                self._set_buffer(mods, n, v)
        self.attach_grads()
        model._flat_state = self
        self.model = model

    @staticmethod
    def _set_buffer(mods, qualified, tensor):
        owner, _, leaf = qualified.rpartition('.')
        mods[owner]._buffers[leaf] = tensor

    @staticmethod
    def of(model):
        """The model's FlatState, created on first use."""
        fs = getattr(model, '_flat_state', None)
        return fs if fs is not None else FlatState(model)

    # ---- gradients -------------------------------------------------------------------
    def attach_grads(self):
        """Point every trainable parameter's ``.grad`` at its slice of the gradient arena, so
        autograd accumulates in place (``zero_grad(set_to_none=True)`` detaches them again).  Runs between the forward
        and the backward pass of every step (the optimizer hook zeroes there), i.e. on the host's critical path: the
        slice addresses are computed once, a parameter that already points at its slice costs one comparison."""
        ptrs = getattr(self, '_grad_ptrs', None)
        base = self.grads.data_ptr()
        if ptrs is None or self._grad_ptr_base != base:
            ptrs = self._grad_ptrs = [base + 4 * seg.offset for seg in self.param_segments]
            self._grad_ptr_base = base
        for seg, p, ptr in zip(self.param_segments, self._params, ptrs):
            if p.requires_grad:
                g = p.grad
                if g is None or g.data_ptr() != ptr:
                    p.grad = _view_as_param(self.grads, seg)
                    p._yv4_grad_in_arena = True      # conv weights: dW may be accumulated here directly (train_ops)

    def zero_grad(self):
        self.grads.zero_()
        self.attach_grads()

    def grads_attached(self):
        for seg, p in zip(self.param_segments, self._params):
            if p.requires_grad and (p.grad is None or
                                    p.grad.data_ptr() != self.grads.data_ptr() + 4 * seg.offset):
                return False
        return True

    def bump_versions(self):
        """Tell torch that kernels (or arena-level copies) rewrote the state: parameters carry
        their own version counters (``p.data = view`` does not share the arena's), and the
        eval-mode plan caches key on them (``HipModule._param_version``)."""
        for p in self._params:
            torch.autograd.graph.increment_version(p)
        torch.autograd.graph.increment_version(self.values)
        torch.autograd.graph.increment_version(self.ints)
        from . import train_ops          # the packed conv operands of the training step are stale too
        train_ops.invalidate_packed_weights()

    # ---- tables for the kernels ------------------------------------------------------------
    def segment_offsets(self):
        """``nseg+1`` float offsets of the parameter segments (the last one = ``n_param``)."""
        return [s.offset for s in self.param_segments] + [self.n_param]

    def param_index(self):
        return {id(p): i for i, p in enumerate(self._params)}

    def __repr__(self):
        return (f'FlatState({len(self.param_segments)} params / {self.n_param} floats, '
                f'{len(self.buffer_segments)} float buffers, {len(self.int_segments)} int buffers, {self.device})')


def flat_zero_grad(model):
    """``model.zero_grad()`` for a re-homed model: one memset, ``.grad`` views kept."""
    fs = getattr(model, '_flat_state', None)
    if fs is None:
        nn.Module.zero_grad(model)
    else:
        fs.zero_grad()
