#!/usr/bin/env python
"""Per-wave s_memtime sums of the fp32 wide 3x3 kernel's K-tile parts and the clock the chip holds in it (diagnostic build only).

  YV4_LIB_PATH=mmdet-yolov4_amd/lib_var/libyv4_w3f_stamp.so python tools/stamp_w3f.py --cin 256 --cout 256 --hw 38

The library: tools/build_src_variants.sh w3f_stamp:conv3x3_wide_f32:-DYV4_W3F_STAMP.  Prints per wave role (waves 0-3 / 4-7) the
cycles spent in: 0 the prologue (first fills, until the first barrier), 1 the weight pieces' issue at the top of a K tile,
2 the four phases (fragment reads, MFMAs, the next group's image pieces), 3 the counted wait, 4 the barrier, 5 tile set-up,
6 epilogue; the matrix cycles a wave issues (its MFMAs x 32) against the span; and the clock = s_memtime span /
s_memrealtime span x 100 MHz (MI355X_MICROARCH.md, DVFS give-back (6))."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mmdet_yolov4_amd as pkg  # noqa: E402
from mmdet_yolov4_amd._lib import ConvDesc  # noqa: E402

NAMES = ['prologue', 'issueW', 'phases', 'vmcnt', 'barrier', 'tile-setup', 'epilogue']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cin', type=int, default=256)
    ap.add_argument('--cout', type=int, default=256)
    ap.add_argument('--hw', type=int, default=38)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--tile', type=int, default=10)
    ap.add_argument('--warm', type=int, default=200, help='launches before the stamped one (the clock settles over ~100 ms)')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    lib = pkg._lib.lib()
    raw = C.CDLL(pkg._lib.LIB_PATH)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    x = torch.randn(a.batch * a.hw * a.hw * a.cin, device=dev)
    w = torch.randn(a.cout * 9 * a.cin, device=dev) * 0.05
    y = torch.empty(a.batch * a.hw * a.hw * a.cout, device=dev)
    sc = torch.ones(a.cout, device=dev)
    sh = torch.zeros(a.cout, device=dev)
    d = ConvDesc()
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = a.batch, a.hw, a.hw, a.cin, a.hw, a.hw, a.cout
    d.KH = d.KW = 3
    d.stride, d.pad = 1, 1
    d.x_cstride, d.y_cstride = a.cin, a.cout
    d.act1 = 1
    d.tile = a.tile

    def launch():
        rc = lib.yv4_conv_bn_act_fwd(C.byref(d), x.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, None, None,
                                     y.data_ptr(), stream)
        assert rc == 0, lib.yv4_last_error()

    for _ in range(a.warm):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(20):
        launch()
    e0.record()
    launch()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    n = 256 * 64
    out = (C.c_ulonglong * n)()
    assert raw.yv4_debug_w3f_stamps(out, n) == 0
    st = np.frombuffer(out, dtype=np.uint64).reshape(256, 8, 8).astype(np.float64)
    live = st[:, 0, :7].sum(-1) > 0
    st = st[live]
    M = a.batch * a.hw * a.hw
    flops = 2.0 * M * a.cout * 9 * a.cin
    print(f'{a.cin}->{a.cout} @{a.hw} batch {a.batch} fp32 tile {a.tile}: {us:.1f} us (stamped build) = {flops / us / 1e6:.1f} TFLOP/s, '
          f'{int(live.sum())} workgroups live')
    span = st[:, :, :7].sum(-1)
    rt = st[:, :, 7]
    clk = span / np.maximum(rt, 1) * 100.0
    print(f'cycles per wave (stamped span): mean {span.mean():.0f}  min {span.min():.0f}  max {span.max():.0f};  clock '
          f'{np.median(clk):.0f} MHz (median over waves; {clk.min():.0f}-{clk.max():.0f});  span = {np.median(rt) / 100:.1f} us of the {us:.1f}')
    # matrix cycles issued by the CU's four SIMDs: every output of the layer once, 16x16x4 MFMA = 1 024 MACs in 32 cycles
    mfma_cycles_cu = flops / 2 / 1024 * 32 / 4 / max(int(live.sum()), 1)
    print(f'matrix cycles per SIMD of a live CU {mfma_cycles_cu:.0f} = {mfma_cycles_cu / span.mean():.3f} of the span')
    for role, sl in (('waves 0-3', slice(0, 4)), ('waves 4-7', slice(4, 8))):
        m = st[:, sl, :7].mean((0, 1))
        print(f'{role}: ' + ' '.join(f'{NAMES[i]}:{m[i]:.0f}' for i in range(7)) + f'  total {m.sum():.0f}')
    m = st[:, :, :7].mean((0, 1))
    print('share of the span: ' + ' '.join(f'{NAMES[i]}:{m[i] / m.sum():.3f}' for i in range(7)))


if __name__ == '__main__':
    main()
