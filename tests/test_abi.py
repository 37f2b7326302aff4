"""The C-ABI library loads and exports every symbol include/yv4.h declares (no compute)."""
import os
import re

import mmdet_yolov4_amd as pkg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, 'include', 'yv4.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return set(re.findall(r'\b(yv4_[a-z0-9_]+)\s*\(', text))


def test_header_binding_library_agree():
    declared = _declared()
    assert declared, 'no declarations parsed'
    assert declared == set(pkg._lib.SIGNATURES), (declared ^ set(pkg._lib.SIGNATURES))
    lib = pkg._lib.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.yv4_abi_version() == pkg._lib.ABI_VERSION
    assert lib.yv4_arch() == b'gfx950'


def test_struct_layout_matches_header():
    import ctypes
    assert ctypes.sizeof(pkg._lib.ConvDesc) == 22 * 4
    assert ctypes.sizeof(pkg._lib.LevelDesc) == 8 + 3 * 4 + 4 + 8 * 4 * 4 or ctypes.sizeof(pkg._lib.LevelDesc) == 8 + 3 * 4 + 8 * 4 * 4 + 4


def test_argument_validation_without_gpu():
    """Entry points reject bad arguments before touching the device."""
    import ctypes
    lib = pkg._lib.lib()
    d = pkg._lib.ConvDesc()
    assert lib.yv4_conv_bn_act_fwd(ctypes.byref(d), None, None, None, None, None, None, None, None, None) == -1
    assert b'null' in lib.yv4_last_error()
    assert lib.yv4_mish_fwd(None, None, 16, 0, None) == -1
    assert lib.yv4_mish_fwd(None, None, 0, 0, None) == 0          # empty tensor: nothing to do
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = 2, 8, 8, 32, 8, 8, 64
    d.KH = d.KW = 3; d.stride = 1; d.pad = 1
    assert lib.yv4_conv_flops(ctypes.byref(d)) == 2.0 * 2 * 8 * 8 * 64 * 9 * 32
    assert lib.yv4_conv_pick_tile(ctypes.byref(d)) in (1, 2, 3, 4, 5, 6, 7)


def test_cpu_tensors_are_refused():
    import pytest
    import torch
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        pkg.mish_forward(torch.zeros(8))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        pkg.batched_nms(torch.zeros(1, 4), torch.zeros(1), torch.zeros(1, dtype=torch.long), dict(iou_threshold=0.5))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        pkg.Conv(4, 8, 3).eval()(torch.zeros(1, 4, 8, 8))
