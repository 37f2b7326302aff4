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
    assert ctypes.sizeof(pkg._lib.ConvDesc) == 23 * 4 and pkg._lib.ConvDesc.flags.offset == 22 * 4     # ABI 7: + flags
    # yv4_loss_level / yv4_loss_desc (gcc sizeof on include/yv4.h: 176 / 1016, slot_anchor at 960, ABI 6's losses at 1008)
    assert ctypes.sizeof(pkg._lib.LossLevel) == 176 and ctypes.sizeof(pkg._lib.LossDesc) == 1016
    assert pkg._lib.LossDesc.slot_anchor.offset == 960 and pkg._lib.LossDesc.losses.offset == 1008
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
    """Everything with a plan or kernel behind it refuses CPU tensors.  (The standalone Mish op is the exception the
    reference makes too: mish.cc:14-22 dispatches on is_cuda -- test_mish_host_loop_matches_the_reference_fixture.)"""
    import pytest
    import torch
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        pkg.batched_nms(torch.zeros(1, 4), torch.zeros(1), torch.zeros(1, dtype=torch.long), dict(iou_threshold=0.5))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        pkg.Conv(4, 8, 3).eval()(torch.zeros(1, 4, 8, 8))


def test_binding_argument_types_match_the_header():
    """Every prototype in include/yv4.h against the ctypes signature bound to it: argument count, and per argument
    the C scalar type (int / int64_t / float / double / size_t) or pointer-ness.  A wrong argtype (an int64_t bound
    as c_int, a missing argument) would corrupt the call silently."""
    import ctypes
    text = open(os.path.join(ROOT, 'include', 'yv4.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    text = re.sub(r'//[^\n]*', '', text)
    protos = re.findall(r'\b([A-Za-z_][\w\s\*]*?)\b(yv4_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;', text, flags=re.S)
    assert len(protos) >= 50
    scalar = {'int': ctypes.c_int, 'int32_t': ctypes.c_int, 'int64_t': ctypes.c_int64, 'float': ctypes.c_float,
              'double': ctypes.c_double, 'size_t': ctypes.c_size_t}

    def is_pointer_type(t):
        return t in (ctypes.c_void_p, ctypes.c_char_p) or isinstance(t, type(ctypes.POINTER(ctypes.c_int)))

    seen = set()
    for ret, name, args in protos:
        if name not in pkg._lib.SIGNATURES:
            continue
        seen.add(name)
        restype, argtypes = pkg._lib.SIGNATURES[name]
        args = ' '.join(args.split())
        params = [] if args in ('', 'void') else [a.strip() for a in args.split(',')]
        assert len(params) == len(argtypes), f'{name}: header has {len(params)} arguments, binding {len(argtypes)}'
        for i, (prm, bound) in enumerate(zip(params, argtypes)):
            if '*' in prm or '[' in prm:
                assert is_pointer_type(bound), f'{name} arg {i} ({prm}): bound as {bound}, header says pointer'
            else:
                ctype = prm.replace('const ', '').split()[0]
                assert ctype in scalar, f'{name} arg {i}: unhandled C type {prm!r}'
                assert bound is scalar[ctype], f'{name} arg {i} ({prm}): bound as {bound}'
        ret = ret.replace('const', '').strip()
        if '*' in ret:
            assert is_pointer_type(restype), name
        elif ret in scalar:
            assert restype is scalar[ret], f'{name}: returns {ret}, bound as {restype}'
    assert seen == set(pkg._lib.SIGNATURES), set(pkg._lib.SIGNATURES) - seen


def test_mish_host_loop_matches_the_reference_fixture(golden):
    """mish.cc:14-33: a CPU tensor runs the op's host loop.  The library's own loop (csrc/elementwise.hip
    yv4_mish_*_host, NOT the oracle) against tests/golden/mish.npz, which the reference's unmodified mish_cpu.cc produced:
    bit for bit in float32 and float64; Half / BFloat16 are refused as the reference's CPU dispatch refuses them."""
    import numpy as np
    import torch
    g = golden('mish')
    x, gr = torch.from_numpy(g['x']), torch.from_numpy(g['g'])
    np.testing.assert_array_equal(pkg.mish_forward(x).numpy(), g['y'])
    np.testing.assert_array_equal(pkg.mish_backward(gr, x).numpy(), g['gin'])
    np.testing.assert_array_equal(pkg.mish_forward(x.double()).numpy(), g['y64'])
    np.testing.assert_array_equal(pkg.mish_backward(gr.double(), x.double()).numpy(), g['gin64'])
    import pytest
    with pytest.raises(RuntimeError, match='not implemented for this dtype'):      # AT_DISPATCH_ALL_TYPES has no Half
        pkg.mish_forward(x.half())
    # the autograd Function on CPU tensors, end to end
    xr = x.clone().requires_grad_(True)
    pkg.MishFunction.apply(xr).backward(gr)
    np.testing.assert_array_equal(xr.grad.numpy(), g['gin'])
