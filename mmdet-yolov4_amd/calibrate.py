"""Data-dependent initialisation of a randomly initialised detector ON THE DEVICE.

There is no network access for checkpoints, so benchmarks run random-init weights
("running stats from one seeded calibration batch").  With BatchNorm at its
initial statistics the activations of a 115-conv network drift by orders of magnitude;
this walks a compiled plan in launch order and, for every conv whose epilogue was folded
from a BatchNorm, measures the per-channel mean/variance of what that BN sees on a
calibration batch (the raw accumulator for stage 1, the stage-1 output for a CSP-level
stage 2), writes them into the module's ``running_mean`` / ``running_var`` and refreshes
the folded scale/shift in place.  It is the eval-mode equivalent of one training-mode
forward pass with momentum 1.  torch is used for the moment reductions (setup time only).
"""
import ctypes as C

import torch

from . import _lib
from ._lib import ConvDesc, check
from .plan import bn_affine


def _copy_desc(d, **over):
    n = ConvDesc()
    C.memmove(C.byref(n), C.byref(d), C.sizeof(ConvDesc))
    for k, v in over.items():
        setattr(n, k, v)
    return n


def _view_tensor(view):
    b = view.buf
    return b.tensor.view(b.N * b.H * b.W, b.C)[:, view.coff:view.coff + view.C]


def _set_stats(bn, lo, hi, x2d):
    var, mean = torch.var_mean(x2d.double(), dim=0, unbiased=False)
    with torch.no_grad():
        bn.running_mean[lo:hi] = mean.float().to(bn.running_mean.device)
        bn.running_var[lo:hi] = var.float().clamp_min(1e-12).to(bn.running_var.device)


@torch.no_grad()
def calibrate_bn(plan, *inputs):
    """Run ``plan`` once on ``inputs`` while fitting every folded BN to the batch."""
    assert plan.finalized and plan.graph is None
    assert not any(op.info.get('fused') for op in plan.ops), \
        'calibrate_bn walks single-conv launches: calibrate through the fp32 plan (the statistics live in the modules) ' \
        'or compile the 16-bit plan with YV4_STEM_FUSE=0'
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for slot, t in zip(plan.inputs, inputs):
        slot['src'] = t.contiguous()
    dev = plan.device
    for op in plan.ops:
        if op.kind != 'conv' or (op.info['bn1'] is None and op.info['bn2'] is None):
            if op.kind not in ('decode', 'nms', 'reset'):
                op.fn(stream)
            continue
        L = op.info['launch']
        d = L['d']
        saved = dict(L)
        Cout = d.Cout
        ones = torch.ones(Cout, device=dev)
        zeros = torch.zeros(Cout, device=dev)
        if op.info['bn1'] is not None:
            L.update(d=_copy_desc(d, act1=0, act2=0), s1=ones, t1=zeros, s2=None, t2=None, res=None)
            op.fn(stream)
            torch.cuda.synchronize()
            # one BatchNorm range, or several behind consecutive output channels (two sibling 1x1 convs in one launch)
            parts = op.info['bn1'] if isinstance(op.info['bn1'], list) else [op.info['bn1']]
            raw = _view_tensor(op.info['out'])
            c0 = 0
            for bn, lo, hi in parts:
                _set_stats(bn, lo, hi, raw[:, c0:c0 + hi - lo])
                s, t = bn_affine(bn)
                saved['s1'][c0:c0 + hi - lo].copy_(s[lo:hi].to(dev))
                saved['t1'][c0:c0 + hi - lo].copy_(t[lo:hi].to(dev))
                c0 += hi - lo
            assert c0 == Cout
        if op.info['bn2'] is not None:
            bn, lo, hi = op.info['bn2']
            L.update(saved)
            L.update(d=_copy_desc(d, act2=0), s2=None, t2=None)
            op.fn(stream)
            torch.cuda.synchronize()
            _set_stats(bn, lo, hi, _view_tensor(op.info['out']))
            s, t = bn_affine(bn)
            saved['s2'].copy_(s[lo:hi].to(dev))
            saved['t2'].copy_(t[lo:hi].to(dev))
        L.update(saved)
        op.fn(stream)
    torch.cuda.synchronize()
