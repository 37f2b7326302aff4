"""torch-tensor front ends of the C-ABI ops (``include/yv4.h``).

PyTorch is plumbing here: it owns device memory and the current HIP stream; every
number is produced by ``libyv4_hip.so``.  All wrappers require CUDA (ROCm) tensors
and raise otherwise -- there is no CPU fallback.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import check

_DT = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16,
       torch.float64: _lib.F64}


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f'{name} must be a torch.Tensor')
    if not t.is_cuda:
        raise RuntimeError(
            f'{name} is on {t.device}: the yv4 ops run on an MI355X only (HIP path, no CPU fallback)')


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


# ---- Mish ---------------------------------------------------------------------------
def mish_forward(input):
    """``mish_cuda_ext.mish_forward`` (mmdet/ops/mish_cuda/src/mish.cc:14-22): new
    tensor ``empty_like(input)``; input must be contiguous.  Like the reference's dispatcher a CPU tensor runs the
    op's host loop (``yv4_mish_fwd_host``: the library's own restatement of mish.h:16-18 -- the op's CPU behaviour, not
    a path of any plan)."""
    if not isinstance(input, torch.Tensor):
        raise TypeError('input must be a torch.Tensor')
    if not input.is_contiguous():
        raise RuntimeError('mish_forward: input must be contiguous')
    if input.dtype not in _DT:
        raise RuntimeError(f'mish_forward: unsupported dtype {input.dtype}')
    out = torch.empty_like(input)
    if not input.is_cuda:
        check(_lib.lib().yv4_mish_fwd_host(_ptr(input), _ptr(out), input.numel(), _DT[input.dtype]), 'yv4_mish_fwd_host')
        return out
    check(_lib.lib().yv4_mish_fwd(_ptr(input), _ptr(out), input.numel(), _DT[input.dtype],
                                  stream_ptr()), 'yv4_mish_fwd')
    return out


def mish_backward(grad_out, input):
    """``mish_cuda_ext.mish_backward`` (mish.cc:24-33); dispatches on ``grad_out.is_cuda`` as the reference does."""
    if not (isinstance(grad_out, torch.Tensor) and isinstance(input, torch.Tensor)):
        raise TypeError('grad_out and input must be torch.Tensors')
    if grad_out.device != input.device:
        raise RuntimeError(f'mish_backward: grad_out is on {grad_out.device}, input on {input.device}')
    if not (grad_out.is_contiguous() and input.is_contiguous()):
        raise RuntimeError('mish_backward: tensors must be contiguous')
    if grad_out.dtype != input.dtype or grad_out.shape != input.shape:
        raise RuntimeError('mish_backward: grad_out and input must match in dtype and shape')
    if input.dtype not in _DT:
        raise RuntimeError(f'mish_backward: unsupported dtype {input.dtype}')
    gin = torch.empty_like(input)
    if not grad_out.is_cuda:
        check(_lib.lib().yv4_mish_bwd_host(_ptr(grad_out), _ptr(input), _ptr(gin), input.numel(), _DT[input.dtype]),
              'yv4_mish_bwd_host')
        return gin
    check(_lib.lib().yv4_mish_bwd(_ptr(grad_out), _ptr(input), _ptr(gin), input.numel(),
                                  _DT[input.dtype], stream_ptr()), 'yv4_mish_bwd')
    return gin


class MishFunction(torch.autograd.Function):
    """``MishCudaFunction`` (mmdet/ops/mish_cuda/mish.py:18-36): saves the input."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda')
    def forward(ctx, inp):
        if not inp.is_contiguous():
            inp = inp.contiguous()
        ctx.save_for_backward(inp)
        return mish_forward(inp)

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, grad_out):
        inp, = ctx.saved_tensors
        if not grad_out.is_contiguous():
            grad_out = grad_out.contiguous()
        if not ctx.needs_input_grad[0]:
            return (None, )
        return mish_backward(grad_out, inp)


# ---- layout -------------------------------------------------------------------------
def nchw_to_nhwc(src, dst, dst_cstride, dst_coff=0, zero_pad=0):
    N, Cc, H, W = src.shape
    check(_lib.lib().yv4_nchw_to_nhwc(_ptr(src), _ptr(dst), N, Cc, H, W, dst_cstride, dst_coff,
                                      zero_pad, stream_ptr()), 'yv4_nchw_to_nhwc')


def nhwc_to_nchw(src, dst, C_, H, W, src_cstride, src_coff=0):
    N = dst.shape[0]
    check(_lib.lib().yv4_nhwc_to_nchw(_ptr(src), _ptr(dst), N, C_, H, W, src_cstride, src_coff,
                                      stream_ptr()), 'yv4_nhwc_to_nchw')


# ---- NMS ----------------------------------------------------------------------------
SPLIT_THR_DEFAULT = 10000


FAST_NMS_CAP = 16384   # candidates one workgroup sorts in LDS (csrc/postproc.hip kSortCap)


def _nms_single(boxes, scores, labels, iou_threshold, max_out, split_thr, class_agnostic=False):
    """One image through yv4_nms_prepare + yv4_nms_images (n < split_thr) or yv4_nms_split.
    Returns (dets(k,5), keep(k,))."""
    n = boxes.shape[0]
    dev = boxes.device
    L = _lib.lib()
    keys = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
    counts = torch.empty(1, dtype=torch.int32, device=dev)
    maxc = torch.empty(1, dtype=torch.float32, device=dev)
    check(L.yv4_nms_prepare(_ptr(boxes), _ptr(scores), n, _ptr(keys), _ptr(counts), _ptr(maxc),
                            stream_ptr()), 'yv4_nms_prepare')
    cap = max_out if max_out > 0 else n
    cap = max(min(cap, n), 1)
    dets = torch.empty((cap, 5), dtype=torch.float32, device=dev)
    olab = torch.empty(cap, dtype=torch.int32, device=dev)
    oidx = torch.empty(cap, dtype=torch.int64, device=dev)
    ocnt = torch.empty(1, dtype=torch.int32, device=dev)
    if n < split_thr:
        if n > FAST_NMS_CAP:
            raise NotImplementedError(
                f'single-call NMS over {n} > {FAST_NMS_CAP} candidates is not built '
                "(mmcv's default split_thr=10000 switches to the per-class path before that)")
        check(L.yv4_nms_images(_ptr(keys), n, _ptr(counts), _ptr(maxc), _ptr(boxes), n,
                               None if class_agnostic else _ptr(labels), n, 0, 1, float(iou_threshold), cap,
                               split_thr, _ptr(dets), _ptr(olab), _ptr(oidx), _ptr(ocnt), stream_ptr()),
              'yv4_nms_images')
    else:
        # mmcv's split branch loops over unique(idxs) even when class_agnostic (then without the
        # coordinate offset): max_coord = -1 makes the offset unit zero.
        work = torch.empty(max(L.yv4_nms_split_work(n), 256), dtype=torch.uint8, device=dev)
        mc = -1.0 if class_agnostic else float(maxc.item())
        check(L.yv4_nms_split(_ptr(keys), n, mc, _ptr(boxes), _ptr(labels), 0,
                              float(iou_threshold), cap, _ptr(work), _ptr(dets), _ptr(olab),
                              _ptr(oidx), _ptr(ocnt), stream_ptr()), 'yv4_nms_split')
    k = int(ocnt.item())
    if k < 0:
        raise RuntimeError('yv4_nms_images flagged the split path unexpectedly')
    return dets[:k], oidx[:k]


def nms(boxes, scores, iou_threshold, offset=0, score_threshold=0, max_num=-1):
    """``mmcv.ops.nms.nms`` (1.3.x signature).  ``offset`` must be 0 (the only value the
    reference's configs use).  Returns ``(dets(k,5), keep(k,) int64)`` ordered by
    descending score, ties by ascending index."""
    assert boxes.size(1) == 4
    assert boxes.size(0) == scores.size(0)
    if offset != 0:
        raise NotImplementedError('nms: only offset=0 is built')
    _need_cuda(boxes, 'boxes')
    boxes = boxes.contiguous().float()
    scores = scores.contiguous().float()
    inds = None
    if score_threshold > 0:
        valid = scores > score_threshold
        inds = valid.nonzero(as_tuple=False).squeeze(1)
        boxes, scores = boxes[inds].contiguous(), scores[inds].contiguous()
    if boxes.shape[0] == 0:
        return boxes.new_zeros((0, 5)), torch.zeros((0,), dtype=torch.int64, device=boxes.device)
    dets, keep = _nms_single(boxes, scores, None, iou_threshold, max_num, 1 << 30, True)
    if inds is not None:
        keep = inds[keep]
    return dets, keep


def set_deterministic(on=True):
    """``yv4_set_deterministic``: every floating-point sum that meets in atomics (BatchNorm statistics and their backward,
    loss sums and row gradients, bias gradients, the SPP scatter, the gradient norm) runs on fixed-point integer words or
    in a fixed order, so a training step gives the same bits run to run (the weight gradient's ordered form is this
    host's default already).  The counterpart of ``torch.use_deterministic_algorithms`` for the reference's step, whose
    BatchNorm (torch.nn.BatchNorm2d via mmdet/models/backbones/darknetcsp.py:15-35) is deterministic.  Process-wide;
    switch it between steps."""
    check(_lib.lib().yv4_set_deterministic(1 if on else 0), 'yv4_set_deterministic')


def deterministic():
    return bool(_lib.lib().yv4_get_deterministic())


def set_nms_iou_form(form):
    """Choose which of mmcv-full 1.3.x's two suppression predicates every NMS launch of this process applies:
    'div' (default) -- ``inter / (Sa + Sb - inter) > thr``, mmcv's CPU kernel and the documented definition of this
    package; 'mul' -- ``inter > thr * (Sa + Sb - inter)``, mmcv's CUDA kernel, i.e. what the reference executes on a GPU.
    They select differently only on pairs whose fp32 IoU rounds across the threshold (tests/golden/nms_boundary.npz).
    Also settable through the environment: ``YV4_NMS_IOU_FORM=mul``."""
    code = {'div': _lib.NMS_IOU_DIV, 'mul': _lib.NMS_IOU_MUL, 0: 0, 1: 1}[form]
    check(_lib.lib().yv4_nms_set_iou_form(code), 'yv4_nms_set_iou_form')


def get_nms_iou_form():
    return {0: 'div', 1: 'mul'}[_lib.lib().yv4_nms_get_iou_form()]


def batched_nms(boxes, scores, idxs, nms_cfg, class_agnostic=False):
    """``mmcv.ops.nms.batched_nms`` as called at
    mmdet/core/post_processing/bbox_nms.py:84.  ``idxs`` may live on the CPU (Q6)."""
    _need_cuda(boxes, 'boxes')
    nms_cfg_ = dict(nms_cfg)
    class_agnostic = nms_cfg_.pop('class_agnostic', class_agnostic)
    nms_type = nms_cfg_.pop('type', 'nms')
    if nms_type != 'nms':
        raise NotImplementedError(f'batched_nms: nms type {nms_type!r} is not built (only "nms")')
    split_thr = nms_cfg_.pop('split_thr', SPLIT_THR_DEFAULT)
    iou_threshold = nms_cfg_.pop('iou_threshold', nms_cfg_.pop('iou_thr', None))
    if iou_threshold is None:
        raise KeyError('batched_nms: nms_cfg needs iou_threshold')
    score_threshold = nms_cfg_.pop('score_threshold', 0)
    max_num = nms_cfg_.pop('max_num', -1)
    if nms_cfg_.pop('offset', 0) != 0:
        raise NotImplementedError('batched_nms: only offset=0 is built')
    if nms_cfg_:
        raise TypeError(f'batched_nms: unexpected nms_cfg keys {sorted(nms_cfg_)}')
    boxes = boxes.contiguous().float()
    scores = scores.contiguous().float()
    n = boxes.shape[0]
    if n == 0:
        return boxes.new_zeros((0, 5)), torch.zeros((0,), dtype=torch.int64, device=boxes.device)
    labels = idxs.to(device=boxes.device, dtype=torch.int32).contiguous()
    inds = None
    if score_threshold > 0:
        inds = (scores > score_threshold).nonzero(as_tuple=False).squeeze(1)
        boxes, scores = boxes[inds].contiguous(), scores[inds].contiguous()
        if labels is not None:
            labels = labels[inds].contiguous()
        if boxes.shape[0] == 0:
            return boxes.new_zeros((0, 5)), torch.zeros((0,), dtype=torch.int64, device=boxes.device)
    dets, keep = _nms_single(boxes, scores, labels, iou_threshold, max_num, split_thr, class_agnostic)
    if inds is not None:
        keep = inds[keep]
    return dets, keep


def multiclass_nms(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1, score_factors=None,
                   return_inds=False):
    """``mmdet/core/post_processing/bbox_nms.py:7-93``; same shapes incl. the empty
    case (Q7: boxes ``(0,4)``, labels int64 ``(0,)``)."""
    num_classes = multi_scores.size(1) - 1
    if multi_bboxes.shape[1] > 4:
        bboxes = multi_bboxes.view(multi_scores.size(0), -1, 4)
    else:
        bboxes = multi_bboxes[:, None].expand(multi_scores.size(0), num_classes, 4)
    scores = multi_scores[:, :-1]
    labels = torch.arange(num_classes, dtype=torch.long)  # on the CPU, as in the reference (Q6)
    labels = labels.view(1, -1).expand_as(scores)
    bboxes = bboxes.reshape(-1, 4)
    scores = scores.reshape(-1)
    labels = labels.reshape(-1)
    valid_mask = scores > score_thr
    if score_factors is not None:
        score_factors = score_factors.view(-1, 1).expand(multi_scores.size(0), num_classes)
        scores = scores * score_factors.reshape(-1)
    inds = valid_mask.nonzero(as_tuple=False).squeeze(1)
    bboxes, scores, labels = bboxes[inds], scores[inds], labels.to(inds.device)[inds]
    if bboxes.numel() == 0:
        if return_inds:
            return bboxes, labels, inds
        return bboxes, labels
    dets, keep = batched_nms(bboxes, scores, labels, nms_cfg)
    if max_num > 0:
        dets = dets[:max_num]
        keep = keep[:max_num]
    if return_inds:
        return dets, labels[keep], keep
    return dets, labels[keep]
