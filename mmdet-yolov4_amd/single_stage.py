"""``SingleStageDetector`` under the reference's registry name.

Mirror of ``mmdet/models/detectors/single_stage.py:35-112`` and the parts of
``mmdet/models/detectors/base.py:111-169`` an inference caller touches
(``forward(return_loss=False)`` -> ``forward_test`` -> ``simple_test``), with
``bbox2result`` of ``mmdet/core/bbox/transforms.py:99-113``.

``simple_test`` replays ONE plan for the whole pipeline -- NCHW->NHWC, 115 fused convs,
SPP, 2 upsamples, decode+filter, NMS -- built once per (batch, height, width) by
``compile``; the host touches the device twice per batch (scale factors in, detections
out).
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.distributed as dist

from .bricks import HipModule, plan_cache_get, plan_cache_put
from .deferred import DeferredLogVars, read_back_later  # noqa: F401
from .plan import Plan
from .registry import DETECTORS, build_backbone, build_head, build_neck
from .yolocsp_head import collect_results, set_scale_factors


def bbox2result(bboxes, labels, num_classes):
    """core/bbox/transforms.py:99-113."""
    if bboxes.shape[0] == 0:
        return [np.zeros((0, 5), dtype=np.float32) for _ in range(num_classes)]
    if isinstance(bboxes, torch.Tensor):
        bboxes = bboxes.detach().cpu().numpy()
        labels = labels.detach().cpu().numpy()
    return [bboxes[labels == i, :] for i in range(num_classes)]


@DETECTORS.register_module()
class SingleStageDetector(HipModule):

    def __init__(self, backbone, neck=None, bbox_head=None, train_cfg=None, test_cfg=None, pretrained=None,
                 init_cfg=None):
        super().__init__(init_cfg)
        if pretrained is not None:
            backbone = dict(backbone, pretrained=pretrained)
        self.backbone = build_backbone(backbone)
        if neck is not None:
            self.neck = build_neck(neck)
        bbox_head = dict(bbox_head, train_cfg=train_cfg, test_cfg=test_cfg)
        self.bbox_head = build_head(bbox_head)
        self.train_cfg = train_cfg
        self.test_cfg = test_cfg
        self.fp16_enabled = False
        self._engines = {}

    @property
    def with_neck(self):
        return hasattr(self, 'neck') and self.neck is not None

    # ---- plan ------------------------------------------------------------------------------
    def emit_feat(self, plan, x):
        feats = self.backbone.emit(plan, x)
        if self.with_neck:
            feats = self.neck.emit(plan, feats)
        return feats

    def emit(self, plan, x):
        """forward_dummy graph: image -> NHWC pred maps."""
        return self.bbox_head.emit(plan, self.emit_feat(plan, x))

    def compile(self, batch, height, width, device='cuda', rescale=True, graph=False, autotune=False, dtype=None):
        """Build (and cache) the end-to-end inference plan for one input geometry.
        graph=True records the launch list into a hipGraph (one replay per call; the
        reference's batch-1 protocol is otherwise bound by ~124 host launches);
        autotune=True times the candidate conv tiles per layer first; dtype: torch.float32 (default,
        the parity dtype), torch.float16 or torch.bfloat16 (``wrap_fp16_model`` sets the default)."""
        dtype = dtype or getattr(self, 'compute_dtype', torch.float32)
        key = (batch, height, width, str(device), bool(rescale), bool(graph), self._param_version(), dtype)
        eng = plan_cache_get(self._engines, key)
        if eng is None:
            plan = Plan(device, dtype)
            # 16-bit plans keep the image fp32 when the backbone starts with the 3x3 stem (its own
            # fp32 kernel, 16-bit output); otherwise the image is converted like any other tensor
            conv0 = next((m for m in self.backbone.modules() if isinstance(m, torch.nn.Conv2d)), None)
            stem32 = (plan.h16 and isinstance(conv0, torch.nn.Conv2d) and conv0.kernel_size == (3, 3)
                      and conv0.stride == (1, 1) and conv0.padding == (1, 1) and conv0.in_channels == 3
                      and conv0.out_channels <= 64 and conv0.out_channels % 8 == 0)
            x = plan.add_input_nchw(batch, 3, height, width, name='img', dtype=torch.float32 if stem32 else None)
            plan.hint_single_consumer(x)             # the image feeds the backbone's first conv and nothing else
            preds = self.emit(plan, x)
            self.bbox_head.emit_postprocess(plan, preds, rescale=rescale)
            plan.pred_views = preds
            plan.finalize()
            if autotune and torch.device(device).type == 'cuda':
                plan.inputs[0]['src'] = torch.zeros((batch, 3, height, width), dtype=torch.float32, device=device)
                plan._launch_all(__import__('ctypes').c_void_p(torch.cuda.current_stream().cuda_stream))
                plan.autotune()
            if graph:
                plan.capture()
            eng = plan
            plan_cache_put(self._engines, key, eng, 6)
        return eng

    # ---- reference API ------------------------------------------------------------------------
    def extract_feat(self, img):
        x = self.backbone(img)
        if self.with_neck:
            x = self.neck(x)
        return x

    def forward_dummy(self, img):
        return self.bbox_head(self.extract_feat(img))

    def simple_test(self, img, img_metas, rescale=False):
        self._check_eval()
        N, _, H, W = img.shape
        plan = self.compile(N, H, W, device=img.device, rescale=rescale, graph=True)
        set_scale_factors(plan.post, img_metas, rescale)
        plan.run(img)
        bbox_list = collect_results(plan.post, with_nms=True, head=self.bbox_head)
        return [bbox2result(d, l, self.bbox_head.num_classes) for d, l in bbox_list]

    def _check_eval(self):
        if self.training:
            raise NotImplementedError('inference entry points need .eval(): they replay launch plans with folded BatchNorm; '
                                      'a module in .train() mode runs the autograd path (forward_train / train_step)')

    def forward_test(self, imgs, img_metas, **kwargs):
        for var, name in [(imgs, 'imgs'), (img_metas, 'img_metas')]:
            if not isinstance(var, list):
                raise TypeError(f'{name} must be a list, but got {type(var)}')
        if len(imgs) != len(img_metas):
            raise ValueError(f'num of augmentations ({len(imgs)}) != num of image meta ({len(img_metas)})')
        if len(imgs) == 1:
            return self.simple_test(imgs[0], img_metas[0], **kwargs)
        raise NotImplementedError('aug_test (TTA) is not built')

    def forward(self, img, img_metas, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(img, img_metas, **kwargs)
        return self.forward_test(img, img_metas, **kwargs)

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None):
        """single_stage.py:51-79: features through the HIP training ops, then the head's losses."""
        x = self.extract_feat(img)
        return self.bbox_head.forward_train(x, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore)

    def _parse_losses(self, losses):
        """detectors/base.py:171-204: total = sum of the entries whose key contains 'loss'; every log variable is the
        mean over ranks, returned as a python float.  The reference issues one all-reduce and one ``.item()`` (a host
        sync) PER log variable (base.py:197-202); here the variables are stacked on the device, exchanged by ONE
        all-reduce and read back by ONE device-to-host copy -- same values, one sync per step, taken when a value is
        first read (``DeferredLogVars``)."""
        if getattr(losses, 'total', None) is not None and not getattr(losses, 'built', True):
            # the fused head's loss matrix (yolocsp_head.FusedLosses): the same sums from the matrix itself
            m = losses.weighted.detach()
            cols = m.sum(0)
            names = (['loss_cls'] if losses.with_cls else []) + ['loss_conf', 'loss_bbox', 'num_gts', 'loss']
            stacked = torch.cat([cols if losses.with_cls else cols[1:], losses.num_gts.detach().float().reshape(1),
                                 losses.total.detach().reshape(1)])
            if dist.is_available() and dist.is_initialized():
                dist.all_reduce(stacked.div_(dist.get_world_size()))
            if stacked.is_cuda:
                return losses.total, read_back_later(stacked, names)
            return losses.total, OrderedDict(zip(names, stacked.tolist()))
        log_vars = OrderedDict()
        for name, value in losses.items():
            if isinstance(value, torch.Tensor):
                log_vars[name] = value.mean()
            elif isinstance(value, list):
                log_vars[name] = sum(v.mean() for v in value)
            else:
                raise TypeError(f'{name} is not a tensor or list of tensors')
        loss = sum(v for k, v in log_vars.items() if 'loss' in k)
        log_vars['loss'] = loss
        stacked = torch.stack([v.detach().float().reshape(()) for v in log_vars.values()])
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(stacked.div_(dist.get_world_size()))
        if stacked.is_cuda:
            # stream-ordered copy into pinned memory; the host waits for it when a value is first read (see
            # DeferredLogVars) -- normally after the backward pass has been queued
            return loss, read_back_later(stacked, list(log_vars))
        for name, value in zip(list(log_vars), stacked.tolist()):
            log_vars[name] = value
        return loss, log_vars

    def train_step(self, data, optimizer):
        """detectors/base.py:206-239."""
        losses = self(**data)
        loss, log_vars = self._parse_losses(losses)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data['img_metas']))
