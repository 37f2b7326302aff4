"""``YOLOV4Neck`` / ``YOLOV5Neck``: PAN necks, registered under the reference's names.

Mirror of ``mmdet/models/necks/yolo_neck_csp.py`` (:11-238 v4, :241-449 v5).  In the
plan, ``F.interpolate(nearest)`` + ``torch.cat`` (:213-219) become one resample launch
writing the upsampled half of the concat buffer while the lateral 1x1 conv writes the
other half directly; the bottom-up ``cat`` (:229) is a strided conv writing half 0 and a
channel-slice copy of the saved top-down tensor into half 1.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import darknetcsp as _D
from . import train_ops as T
from .bricks import HipModule
from .darknetcsp import BottleneckCSP, BottleneckCSP2, Conv
from .registry import NECKS


class _PANBase(HipModule):

    def _setup_levels(self, in_channels, out_channels, num_outs, start_level, end_level, upsample_cfg):
        assert isinstance(in_channels, list)
        self.in_channels = in_channels
        if isinstance(out_channels, list):
            self.out_channels = out_channels
            num_outs = len(out_channels)
        else:
            assert num_outs is not None
            self.out_channels = [out_channels] * num_outs
        self.num_ins = len(in_channels)
        self.num_outs = num_outs
        self.fp16_enabled = False
        self.upsample_cfg = dict(upsample_cfg)
        if self.upsample_cfg.get('mode', 'nearest') != 'nearest':
            raise NotImplementedError('only nearest upsampling has a kernel (the configs use nearest)')
        if end_level == -1:
            self.backbone_end_level = self.num_ins
            assert num_outs == self.num_ins - start_level
        else:
            self.backbone_end_level = end_level
            assert end_level <= len(in_channels)
            assert num_outs == end_level - start_level
        self.start_level = start_level
        self.end_level = end_level

    def _up_size(self, x, bottom):
        if 'scale_factor' in self.upsample_cfg:
            sf = self.upsample_cfg['scale_factor']
            return int(x.H * sf), int(x.W * sf)
        return bottom.H, bottom.W

    def _up(self, x, bottom):
        if 'scale_factor' in self.upsample_cfg:
            return F.interpolate(x, **self.upsample_cfg)
        return F.interpolate(x, size=bottom.shape[2:], **self.upsample_cfg)

    def _up_target(self, x, bottom):
        """(H, W) ``_up`` produces -- and whether it is the plain nearest mode the resample kernels implement."""
        if self.upsample_cfg.get('mode', 'nearest') != 'nearest':
            return (-1, -1)
        if 'scale_factor' in self.upsample_cfg:
            sf = self.upsample_cfg['scale_factor']
            return int(x.shape[2] * sf), int(x.shape[3] * sf)
        return int(bottom.shape[2]), int(bottom.shape[3])

    def forward(self, inputs):
        return self._dispatch((tuple(inputs),), 'tuple')


@NECKS.register_module()
class YOLOV4Neck(_PANBase):
    """yolo_neck_csp.py:11-238."""

    def __init__(self, in_channels, out_channels, num_outs=None, csp_repetition=3, start_level=0, end_level=-1,
                 norm_cfg=dict(type='BN', requires_grad=True, eps=0.001, momentum=0.03),
                 act_cfg=dict(type='Mish'), csp_act_cfg=dict(type='Mish'),
                 upsample_cfg=dict(mode='nearest'), init_cfg=None):
        if init_cfg is None:
            init_cfg = [dict(type='Xavier', distribution='uniform', layer='Conv2d'),
                        dict(type='Constant', val=1, layer=['_BatchNorm', 'GroupNorm'])]
        super().__init__(init_cfg)
        self._setup_levels(in_channels, out_channels, num_outs, start_level, end_level, upsample_cfg)
        num_outs = self.num_outs
        cfg = dict(norm_cfg=norm_cfg, act_cfg=act_cfg, csp_act_cfg=csp_act_cfg, init_cfg=init_cfg)
        self.pre_upsample_convs = nn.ModuleList()
        self.backbone_pre_concat_convs = nn.ModuleList()
        self.post_upsample_concat_csp = nn.ModuleList()
        self.downsample_convs = nn.ModuleList()
        self.post_downsample_concat_csp = nn.ModuleList()
        self.out_convs = nn.ModuleList()

        # top-down: 1x1 reduce -> upsample -> cat with the reduced lateral -> CSP2
        current = in_channels[self.backbone_end_level - 1]
        to_bottom_up = []
        for i in range(self.backbone_end_level - 1, self.start_level, -1):
            bottom = in_channels[i - 1]
            target = bottom // 2
            to_bottom_up.append(current)
            self.pre_upsample_convs.insert(0, Conv(in_channels=current, out_channels=target, kernel_size=1, **cfg))
            self.backbone_pre_concat_convs.insert(
                0, Conv(in_channels=bottom, out_channels=target, kernel_size=1, **cfg))
            self.post_upsample_concat_csp.insert(
                0, BottleneckCSP2(in_channels=2 * target, out_channels=target, repetition=csp_repetition,
                                  shortcut=False, **cfg))
            current = target

        # bottom-up: 3x3 s2 -> cat with the saved top-down input -> CSP2
        to_output = [current]
        for i in range(self.start_level, self.backbone_end_level - 1):
            top = to_bottom_up.pop(-1)
            self.downsample_convs.append(
                Conv(in_channels=current, out_channels=top, kernel_size=3, stride=2, padding=1, **cfg))
            self.post_downsample_concat_csp.append(
                BottleneckCSP2(in_channels=2 * top, out_channels=top, repetition=csp_repetition,
                               shortcut=False, **cfg))
            to_output.append(top)
            current = top

        for i in range(num_outs):
            self.out_convs.append(Conv(in_channels=to_output[i], out_channels=self.out_channels[i],
                                       kernel_size=3, **cfg))

    def emit(self, plan, inputs):
        assert len(inputs) == len(self.in_channels)
        used = self.backbone_end_level - self.start_level
        x = inputs[self.backbone_end_level - 1]
        merge = []
        for i in range(used - 1, 0, -1):
            lateral_in = inputs[self.start_level + i - 1]
            pre_up = self.pre_upsample_convs[i - 1]
            pre_cat = self.backbone_pre_concat_convs[i - 1]
            csp = self.post_upsample_concat_csp[i - 1]
            t = pre_cat.out_channels
            merge.append(x)
            uh, uw = self._up_size(x, lateral_in)
            assert (uh, uw) == (lateral_in.H, lateral_in.W), 'upsampled map must match the lateral for the concat'
            cat = plan.new_buf(x.N, lateral_in.H, lateral_in.W, 2 * t, 'pan_up_cat')
            pre_cat.emit(plan, lateral_in, out=cat.slice(0, t))           # cat((inputs_bottom, x_up))
            low = pre_up.emit(plan, x)
            plan.resample(low, cat.slice(t, t), name='upsample_nearest')
            x = csp.emit(plan, cat)
        outs = [x]
        for i in range(used - 1):
            down = self.downsample_convs[i]
            csp = self.post_downsample_concat_csp[i]
            top = merge.pop(-1)
            c = down.out_channels
            ho = (x.H + 2 * 1 - 3) // 2 + 1
            wo = (x.W + 2 * 1 - 3) // 2 + 1
            assert (ho, wo) == (top.H, top.W) and top.C == c
            cat = plan.new_buf(x.N, ho, wo, 2 * c, 'pan_down_cat')
            down.emit(plan, x, out=cat.slice(0, c))                       # cat((x_down, saved))
            plan.resample(top, cat.slice(c, c), name='concat_copy')
            x = csp.emit(plan, cat)
            outs.append(x)
        return tuple(self.out_convs[i].emit(plan, outs[i]) for i in range(len(outs)))

    def fwd(self, inputs):
        assert len(inputs) == len(self.in_channels)
        used = self.backbone_end_level - self.start_level
        x = inputs[self.backbone_end_level - 1]
        merge = []
        for i in range(used - 1, 0, -1):
            lateral = self.backbone_pre_concat_convs[i - 1]
            src = inputs[self.start_level + i - 1]
            merge.append(x)
            up = self.pre_upsample_convs[i - 1].fwd(x)
            size = self._up_target(up, src)
            t = lateral.out_channels
            if _D._CAT_SLOTS and lateral.with_norm and lateral.stride == 1 and up.shape[1] == t and T.resample_into_ok(up, size) \
                    and tuple(size) == tuple(src.shape[2:]):
                # both halves of cat((bottom, up)) are written where they belong: the lateral conv's activation by its
                # BN + act pass, the upsampled map by the resample launch (train_ops.CatSlot)
                z = lateral.fwd(src, cat=T.CatSlot(2 * t, 0))
                z = T.resample_into(up, size, T.CatSlot(2 * t, t, z))
            else:
                bottom = lateral.fwd(src)
                z = torch.cat((bottom, self._up(up, bottom)), dim=1)
            x = self.post_upsample_concat_csp[i - 1].fwd(z)
        outs = [x]
        for i in range(used - 1):
            down = self.downsample_convs[i]
            saved = merge.pop(-1)
            c = down.out_channels
            if _D._CAT_SLOTS and down.with_norm and saved.shape[1] == c and T.resample_into_ok(saved, saved.shape[2:]):
                z = down.fwd(x, cat=T.CatSlot(2 * c, 0))
                if tuple(z.shape[2:]) == tuple(saved.shape[2:]) and z.dtype == saved.dtype:
                    z = T.resample_into(saved, saved.shape[2:], T.CatSlot(2 * c, c, z))     # copy into its half
                else:
                    z = torch.cat((z[:, :c], saved), dim=1)
            else:
                z = torch.cat((down.fwd(x), saved), dim=1)
            x = self.post_downsample_concat_csp[i].fwd(z)
            outs.append(x)
        return tuple(self.out_convs[i].fwd(outs[i]) for i in range(len(outs)))


@NECKS.register_module()
class YOLOV5Neck(_PANBase):
    """yolo_neck_csp.py:241-449: like v4 but the 1x1 reduce output itself is what the
    bottom-up path concatenates, CSP blocks are ``BottleneckCSP`` and there are no
    lateral 1x1 convs and no output convs."""

    def __init__(self, in_channels, out_channels, num_outs=None, csp_repetition=3, start_level=0, end_level=-1,
                 norm_cfg=dict(type='BN', requires_grad=True, eps=0.001, momentum=0.03),
                 act_cfg=dict(type='Mish'), csp_act_cfg=dict(type='Mish'),
                 upsample_cfg=dict(mode='nearest'), init_cfg=None):
        if init_cfg is None:
            init_cfg = [dict(type='Xavier', distribution='uniform', layer='Conv2d'),
                        dict(type='Constant', val=1, layer=['_BatchNorm', 'GroupNorm'])]
        super().__init__(init_cfg)
        self._setup_levels(in_channels, out_channels, num_outs, start_level, end_level, upsample_cfg)
        cfg = dict(norm_cfg=norm_cfg, act_cfg=act_cfg, csp_act_cfg=csp_act_cfg, init_cfg=init_cfg)
        self.pre_upsample_convs = nn.ModuleList()
        self.post_upsample_concat_csp = nn.ModuleList()
        self.downsample_convs = nn.ModuleList()
        self.post_downsample_concat_csp = nn.ModuleList()

        current = in_channels[self.backbone_end_level - 1]
        to_bottom_up = []
        for i in range(self.backbone_end_level - 1, self.start_level, -1):
            bottom = in_channels[i - 1]
            target = bottom
            self.pre_upsample_convs.insert(0, Conv(in_channels=current, out_channels=target, kernel_size=1, **cfg))
            to_bottom_up.append(target)
            self.post_upsample_concat_csp.insert(
                0, BottleneckCSP(in_channels=2 * target, out_channels=target, repetition=csp_repetition,
                                 shortcut=False, **cfg))
            current = target

        for i in range(self.start_level, self.backbone_end_level - 1):
            top = to_bottom_up.pop(-1)
            self.downsample_convs.append(
                Conv(in_channels=current, out_channels=top, kernel_size=3, stride=2, padding=1, **cfg))
            out_c = self.out_channels[i - self.start_level + 1]
            self.post_downsample_concat_csp.append(
                BottleneckCSP(in_channels=2 * top, out_channels=out_c, repetition=csp_repetition,
                              shortcut=False, **cfg))
            current = out_c

    def emit(self, plan, inputs):
        assert len(inputs) == len(self.in_channels)
        used = self.backbone_end_level - self.start_level
        x = inputs[self.backbone_end_level - 1]
        merge = []
        for i in range(used - 1, 0, -1):
            lateral = inputs[self.start_level + i - 1]
            pre_up = self.pre_upsample_convs[i - 1]
            csp = self.post_upsample_concat_csp[i - 1]
            t = pre_up.out_channels
            assert lateral.C == t
            x = pre_up.emit(plan, x)
            merge.append(x)
            uh, uw = self._up_size(x, lateral)
            assert (uh, uw) == (lateral.H, lateral.W)
            cat = plan.new_buf(x.N, lateral.H, lateral.W, 2 * t, 'pan5_up_cat')
            plan.resample(lateral, cat.slice(0, t), name='concat_copy')    # cat((inputs_bottom, x_up))
            plan.resample(x, cat.slice(t, t), name='upsample_nearest')
            x = csp.emit(plan, cat)
        outs = [x]
        for i in range(used - 1):
            down = self.downsample_convs[i]
            csp = self.post_downsample_concat_csp[i]
            top = merge.pop(-1)
            c = down.out_channels
            ho = (x.H + 2 - 3) // 2 + 1
            wo = (x.W + 2 - 3) // 2 + 1
            assert (ho, wo) == (top.H, top.W) and top.C == c
            cat = plan.new_buf(x.N, ho, wo, 2 * c, 'pan5_down_cat')
            down.emit(plan, x, out=cat.slice(0, c))
            plan.resample(top, cat.slice(c, c), name='concat_copy')
            x = csp.emit(plan, cat)
            outs.append(x)
        return tuple(outs)

    def fwd(self, inputs):
        assert len(inputs) == len(self.in_channels)
        used = self.backbone_end_level - self.start_level
        x = inputs[self.backbone_end_level - 1]
        merge = []
        for i in range(used - 1, 0, -1):
            bottom = inputs[self.start_level + i - 1]
            x = self.pre_upsample_convs[i - 1].fwd(x)
            merge.append(x)
            size = self._up_target(x, bottom)
            t = x.shape[1]
            if _D._CAT_SLOTS and bottom.shape[1] == t and bottom.dtype == x.dtype and T.resample_into_ok(x, size) \
                    and T.resample_into_ok(bottom, bottom.shape[2:]) and tuple(size) == tuple(bottom.shape[2:]):
                z = T.resample_into(bottom, bottom.shape[2:], T.CatSlot(2 * t, 0))      # copy + upsample, no third tensor
                z = T.resample_into(x, size, T.CatSlot(2 * t, t, z))
            else:
                z = torch.cat((bottom, self._up(x, bottom)), dim=1)
            x = self.post_upsample_concat_csp[i - 1].fwd(z)
        outs = [x]
        for i in range(used - 1):
            down = self.downsample_convs[i]
            saved = merge.pop(-1)
            c = down.out_channels
            z = None
            if _D._CAT_SLOTS and down.with_norm and saved.shape[1] == c and T.resample_into_ok(saved, saved.shape[2:]):
                z = down.fwd(x, cat=T.CatSlot(2 * c, 0))
                if tuple(z.shape[2:]) == tuple(saved.shape[2:]) and z.dtype == saved.dtype:
                    z = T.resample_into(saved, saved.shape[2:], T.CatSlot(2 * c, c, z))
                else:
                    z = torch.cat((z[:, :c], saved), dim=1)
            if z is None:
                z = torch.cat((down.fwd(x), saved), dim=1)
            x = self.post_downsample_concat_csp[i].fwd(z)
            outs.append(x)
        return tuple(outs)
