"""ctypes binding of ``libyv4_hip.so`` (the C-ABI declared in ``include/yv4.h``).

The library is the product: there is no CPU fallback.  ``lib()`` raises
``RuntimeError`` when the shared object has not been built, and every wrapper
raises on a non-zero status with the library's own error message.
"""
import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, 'lib')
LIB_PATH = os.environ.get('YV4_LIB_PATH') or os.path.join(LIB_DIR, 'libyv4_hip.so')   # override: A/B measurement builds
CSRC_DIR = os.path.join(_HERE, 'csrc')

# ---- constants mirrored from include/yv4.h -------------------------------------
ABI_VERSION = 7
STATS_REPLICAS = 64        # YV4_STATS_REPLICAS
GRAD_PREPARE_MAX_WG = 2048  # YV4_GRAD_PREPARE_MAX_WG
F32, F16, BF16, F64 = 0, 1, 2, 3
ACT_NONE, ACT_MISH, ACT_LEAKY, ACT_SWISH = 0, 1, 2, 3
NMS_IOU_DIV, NMS_IOU_MUL = 0, 1
TILE_AUTO, TILE_128x128, TILE_128x64, TILE_64x64, TILE_64x128 = 0, 1, 2, 3, 4
TILE_DMA_64x64, TILE_DMA_128x64, TILE_DMA_128x128, TILE_STEM, TILE_WS_1x1 = 5, 6, 7, 8, 9
HTILE_NAMES = {1: 'h16_128x128', 2: 'h16_128x64', 3: 'h16_64x64', 4: 'h16_pp3x3', 5: 'h16_w3x3', 6: 'h16_ws_1x1', 7: 'h16_s3x3', 8: 'h16_wide'}
# the pinned workgroup-tile shapes YV4_HTILE_W3x3_SHAPE(i) = 13, 21, .. 61 (i = 0 .. 6) and YV4_HTILE_WIDE_SHAPE(i) = 16, 24, .. 48 (tests,
# tile sweeps): same kernels, same names
HTILE_NAMES.update({5 + 8 * (i + 1): 'h16_w3x3' for i in range(7)})
HTILE_NAMES.update({8 + 8 * (i + 1): 'h16_wide' for i in range(5)})
TILE_NAMES = {0: 'auto', 1: '128x128', 2: '128x64', 3: '64x64', 4: '64x128', 5: 'dma64x64', 6: 'dma128x64',
              7: 'dma128x128', 8: 'stem3x3', 9: 'ws_1x1', 10: 'w3x3', 26: 'w3x3', 42: 'w3x3', 58: 'w3x3', 74: 'w3x3', 90: 'w3x3',
              11: 'wide', 27: 'wide', 43: 'wide', 59: 'wide', 75: 'wide', 91: 'wide'}
TILE_W3x3 = 10
CONV_NT_OUT = 1            # yv4_conv_desc.flags (ABI 7): non-temporal output stores
TILE_WIDE = 11


class ConvDesc(C.Structure):
    """``yv4_conv_desc``."""
    _fields_ = [(n, C.c_int32) for n in (
        'N', 'H', 'W', 'Cin', 'Ho', 'Wo', 'Cout', 'KH', 'KW', 'stride', 'pad',
        'x_cstride', 'x_coff', 'y_cstride', 'y_coff', 'r_cstride', 'r_coff',
        'act1', 'act2')] + [('slope1', C.c_float), ('slope2', C.c_float),
                            ('tile', C.c_int32), ('flags', C.c_int32)]


class LevelDesc(C.Structure):
    """``yv4_level_desc``."""
    _fields_ = [('pred', C.c_void_p), ('H', C.c_int32), ('W', C.c_int32),
                ('stride', C.c_int32), ('base_anchors', (C.c_float * 4) * 8)]


class LossLevel(C.Structure):
    """``yv4_loss_level``."""
    _fields_ = [('raw', C.c_void_p), ('draw', C.c_void_p), ('bias', C.c_void_p), ('dbias', C.c_void_p),
                ('H', C.c_int32), ('W', C.c_int32), ('Cp', C.c_int32), ('stride', C.c_int32),
                ('base_anchors', (C.c_float * 4) * 8)]


class LossDesc(C.Structure):
    """``yv4_loss_desc``."""
    _fields_ = [('levels', LossLevel * 5),
                ('num_levels', C.c_int32), ('N', C.c_int32), ('A', C.c_int32), ('num_classes', C.c_int32),
                ('G', C.c_int32), ('dtype', C.c_int32),
                ('gt', C.c_void_p), ('gt_label', C.c_void_p), ('gt_img', C.c_void_p),
                ('shape_thr', C.c_float), ('smooth', C.c_float), ('ratio', C.c_float), ('eps', C.c_float),
                ('w_cls', C.c_float), ('w_conf', C.c_float), ('w_bbox', C.c_float),
                ('slot_anchor', C.c_void_p), ('winner', C.c_void_p), ('npos', C.c_void_p), ('conf_t', C.c_void_p),
                ('gpos', C.c_void_p), ('sums', C.c_void_p), ('losses', C.c_void_p)]


class AugImage(C.Structure):
    """``yv4_aug_image``."""
    _fields_ = [('src', C.c_void_p * 4), ('sh', C.c_int32 * 4), ('sw', C.c_int32 * 4), ('pitch', C.c_int32 * 4),
                ('rh', C.c_int32 * 4), ('rw', C.c_int32 * 4)] + \
               [(n, C.c_int32) for n in ('cxy', 'left', 'top', 'x1', 'y1', 'C', 'S', 'o', 'flip', 'hsv_on')] + \
               [('lut', (C.c_uint8 * 256) * 3)]


class PackDesc(C.Structure):
    """yv4_pack_desc (include/yv4.h)."""
    _fields_ = [('w', C.c_void_p), ('s_co', C.c_int64), ('s_ci', C.c_int64), ('s_kh', C.c_int64), ('s_kw', C.c_int64),
                ('dst', C.c_void_p)] + [(k, C.c_int32) for k in (
                    'Cout', 'Cin', 'KHo', 'KWo', 'kh0', 'kh_step', 'kw0', 'kw_step', 'transpose', 'pad_to', 'dtype',
                    'first_block', 'nblocks', 'rows_per_block')]


_vp, _i, _i64, _f, _sz, _d = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t, C.c_double

#: every exported symbol: name -> (restype, argtypes).  tests/test_abi.py checks
#: this table against include/yv4.h and against the built library.
SIGNATURES = {
    'yv4_abi_version': (C.c_int, []),
    'yv4_last_error': (C.c_char_p, []),
    'yv4_arch': (C.c_char_p, []),
    'yv4_mish_fwd': (C.c_int, [_vp, _vp, _sz, _i, _vp]),
    'yv4_mish_bwd': (C.c_int, [_vp, _vp, _vp, _sz, _i, _vp]),
    'yv4_mish_fwd_host': (C.c_int, [_vp, _vp, _sz, _i]),
    'yv4_mish_bwd_host': (C.c_int, [_vp, _vp, _vp, _sz, _i]),
    'yv4_nchw_to_nhwc': (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'yv4_nhwc_to_nchw': (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'yv4_conv_bn_act_fwd': (C.c_int, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp,
                                      _vp, _vp, _vp, _vp, _vp]),
    'yv4_conv_splitk_workspace': (_sz, [C.POINTER(ConvDesc), C.POINTER(C.c_int)]),
    'yv4_conv_bn_act_fwd_splitk': (C.c_int, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'yv4_conv_h16_splitk_workspace': (_sz, [C.POINTER(ConvDesc), C.POINTER(C.c_int)]),
    'yv4_conv_bn_act_fwd_h16_splitk': (C.c_int, [C.POINTER(ConvDesc), C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'yv4_conv_flops': (C.c_double, [C.POINTER(ConvDesc)]),
    'yv4_conv_pick_tile': (C.c_int, [C.POINTER(ConvDesc)]),
    'yv4_spp_pool_fwd': (C.c_int, [_vp, _i, _i, _i, _i, _i, _i, _vp]),
    'yv4_resample_nearest_fwd': (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _i,
                                           _i, _i, _i, _i, _vp]),
    'yv4_resample_nearest_bwd': (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'yv4_decode_reset': (C.c_int, [_vp, _vp, _i, _vp]),
    'yv4_decode_filter': (C.c_int, [C.POINTER(LevelDesc), _i, _i, _i, _i, _f,
                                    _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp,
                                    _vp, _vp]),
    'yv4_decode_filter_v3': (C.c_int, [C.POINTER(LevelDesc), _i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _i64, _vp,
                                       _vp, _vp, _vp]),
    'yv4_conf_topk_levels_work': (_sz, [_i, _i64, _i]),
    'yv4_conf_topk_levels': (C.c_int, [C.POINTER(LevelDesc), _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    'yv4_conf_topk_work': (_sz, [_i, _i64]),
    'yv4_conf_topk': (C.c_int, [C.POINTER(LevelDesc), _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    'yv4_nms_images': (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _i,
                                 _i, _f, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'yv4_nms_split_work': (_sz, [_i64]),
    'yv4_nms_split': (C.c_int, [_vp, _i64, _f, _vp, _vp, _i, _f, _i, _vp, _vp,
                                _vp, _vp, _vp, _vp]),
    'yv4_nms_prepare': (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    'yv4_nms_set_iou_form': (C.c_int, [_i]),
    'yv4_set_deterministic': (C.c_int, [_i]),
    'yv4_get_deterministic': (C.c_int, []),
    'yv4_conv_stats_fold': (C.c_int, [_vp, _i, _i, _vp, _vp]),
    'yv4_nms_get_iou_form': (C.c_int, []),
    'yv4_conv_wgrad': (C.c_int, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp]),
    'yv4_conv_wgrad_workspace': (_sz, [C.POINTER(ConvDesc), _i]),
    'yv4_conv_wgrad_det': (C.c_int, [C.POINTER(ConvDesc), _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    'yv4_dilate2_fwd': (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'yv4_bn_train_stats': (C.c_int, [_vp, _i64, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    'yv4_bn_act_fwd': (C.c_int, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i64, _i, _i, _f, _vp]),
    'yv4_bn_act_bwd': (C.c_int, [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i64, _i,
                                 _i, _f, _vp]),
    'yv4_conv_bn_act_fwd_h16': (C.c_int, [C.POINTER(ConvDesc), _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'yv4_conv_stem_fwd': (C.c_int, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    'yv4_conv_h16_pick_tile': (C.c_int, [C.POINTER(ConvDesc)]),
    'yv4_stem_down_fwd_h16': (C.c_int, [_i, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, _i, _i, _f, _vp, _i, _i,
                                        _vp]),
    'yv4_nchw_to_nhwc_h16': (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'yv4_nhwc_to_nchw_h16': (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'yv4_spp_pool_fwd_h16': (C.c_int, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'yv4_conv_wgrad_h16': (C.c_int, [C.POINTER(ConvDesc), _i, _vp, _vp, _vp, _vp]),
    'yv4_bn_train_stats_h16': (C.c_int, [_vp, _i, _i64, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    'yv4_bn_act_fwd_h16': (C.c_int, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i64, _i, _i, _f,
                                     _vp]),
    'yv4_bn_act_bwd_h16': (C.c_int, [_vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i64,
                                     _i, _i, _f, _vp]),
    'yv4_conv_scatter_fwd': (C.c_int, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'yv4_conv_scatter_fwd_h16': (C.c_int, [C.POINTER(ConvDesc), _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'yv4_bn_eval_act_bwd': (C.c_int, [_vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i64,
                                      _i, _i, _f, _vp]),
    'yv4_spp_pool_bwd': (C.c_int, [_vp, _i, _i, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    'yv4_bn_act_bwd_accum': (C.c_int, [_vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i64, _i, _i, _f, _i, _vp]),
    'yv4_bn_partial_sums': (C.c_int, [_vp, _i, _i64, _i, _i, _i, _vp, _vp]),
    'yv4_bn_finalize': (C.c_int, [_vp, _i, _i64, _vp, _i, _f, _f, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    'yv4_conv_fwd_stats': (C.c_int, [C.POINTER(ConvDesc), _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    'yv4_bn_act_bwd_sums': (C.c_int, [_vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _f, _vp]),
    'yv4_bn_act_bwd_apply': (C.c_int, [_vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i64, _i64, _vp, _i, _i, _f, _vp]),
    'yv4_pack_weight': (C.c_int, [_vp, _i64, _i64, _i64, _i64] + [_i] * 12 + [_vp, _i, _vp]),
    'yv4_pack_weights_multi': (C.c_int, [_vp, _i, _i, _vp]),
    'yv4_yolo_loss_fwd': (C.c_int, [C.POINTER(LossDesc), _vp]),
    'yv4_yolo_loss_bwd': (C.c_int, [C.POINTER(LossDesc), _vp, _vp]),
    'yv4_mosaic_augment_u8': (C.c_int, [_vp, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    'yv4_augment_boxes': (C.c_int, [_vp, _i, _i, _vp, _vp, _vp, _vp, _i, _d, _d, _f, _f, _vp, _vp, _vp, _vp]),
    'yv4_letterbox_u8': (C.c_int, [_vp, _i, _i, _i, _vp, _i, _i, _i64, _i, _i, _vp, _vp, _i, _i, _i, _vp]),
    'yv4_iou_coco_batched': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _vp, _vp]),
    'yv4_match_coco_batched': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    'yv4_grad_prepare': (C.c_int, [_vp, _i64, _vp, _f, _vp, _vp, _vp]),
    'yv4_sgd_step': (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _i, _vp, _vp]),
    'yv4_loss_scale_update': (C.c_int, [_vp, _vp, _f, _f, _i, _vp]),
    'yv4_ema_update': (C.c_int, [_vp, _vp, _i64, _f, _vp]),
}

_lock = threading.Lock()
_lib = None


def build(verbose=False):
    """Compile the HIP sources for gfx950 into ``lib/libyv4_hip.so`` (in-tree)."""
    cmd = ['make', '-C', CSRC_DIR, '-j4']
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError('building libyv4_hip.so failed (see output above)')
    return LIB_PATH


def lib():
    """Load (once) and return the bound library.  No fallback: raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} is missing: the HIP extension has not been built. '
                'Run `python -c "import __graft_entry__ as g; g.build()"` (or '
                f'`make -C {CSRC_DIR}`). There is no CPU fallback for this path.')
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if a symbol is missing
            fn.restype = res
            fn.argtypes = args
        got = handle.yv4_abi_version()
        if got != ABI_VERSION and not (os.environ.get('YV4_LIB_PATH') and os.environ.get('YV4_LIB_ABI_ANY') == '1'
                                       and got == ABI_VERSION - 1):
            # (YV4_LIB_ABI_ANY=1 with YV4_LIB_PATH: same-box A/B against the previous ABI's build -- ABI 7 only APPENDED
            # yv4_conv_desc.flags, which an ABI-6 library never reads)
            raise RuntimeError(f'libyv4_hip.so ABI {got} != binding ABI {ABI_VERSION}; rebuild')
        form = os.environ.get('YV4_NMS_IOU_FORM', '').lower()
        if form in ('mul', '1', 'product', 'cuda'):
            handle.yv4_nms_set_iou_form(NMS_IOU_MUL)      # mmcv's CUDA-kernel predicate instead of its CPU kernel's
        _lib = handle
    return _lib


class Yv4Error(RuntimeError):
    """A C-ABI call returned a non-zero status."""


def check(status, what):
    if status != 0:
        msg = lib().yv4_last_error().decode('utf-8', 'replace')
        raise Yv4Error(f'{what} failed with status {status}: {msg}')
