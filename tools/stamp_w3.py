#!/usr/bin/env python
"""Per-wave s_memtime sums of the wide 3x3 16-bit kernel's K-tile parts (diagnostic build only).

  YV4_LIB_PATH=mmdet-yolov4_amd/lib_var/libyv4_w3stamp_r1.so python tools/stamp_w3.py --cin 256 --cout 256 --hw 38 --tile 21

The library must have been built from conv3x3_wide_h16.hip with -DYV4_W3_STAMP (tools/stamp_w3.sh).  Prints, per wave
role (waves 0-3 / 4-7), the cycles spent in: 0 the prologue (first fills, until the first barrier), 1 the LOAD intervals (LDS-DMA
issue, fragment reads, waves 4-7: counted wait), 2 the barriers that end them, 3 the MFMA intervals' issue, 4 waves 0-3:
counted wait, 5 the barriers that end them, and per output tile: 6 tile set-up / masks / the groups' re-alignment, 7 epilogue.
--res adds a residual, --flush-mb N overwrites N MB before the launch (300: every cache forgets the layer)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mmdet_yolov4_amd as pkg  # noqa: E402
from mmdet_yolov4_amd._lib import ConvDesc  # noqa: E402

NAMES = ['prologue', 'load', 'bar(load)', 'mfma', 'vmcnt', 'bar(mfma)', 'tile-setup', 'epilogue']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cin', type=int, default=256)
    ap.add_argument('--cout', type=int, default=256)
    ap.add_argument('--hw', type=int, default=38)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--tile', type=int, default=5)
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--res', action='store_true')
    ap.add_argument('--flush-mb', type=int, default=0)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    lib = pkg._lib.lib()
    raw = C.CDLL(pkg._lib.LIB_PATH)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    tdt = torch.bfloat16 if a.dtype == 'bf16' else torch.float16
    code = 2 if a.dtype == 'bf16' else 1
    x = torch.randn(a.batch * a.hw * a.hw * a.cin, device=dev).to(tdt)
    w = (torch.randn(a.cout * 9 * a.cin, device=dev) * 0.05).to(tdt)
    y = torch.empty(a.batch * a.hw * a.hw * a.cout, device=dev, dtype=tdt)
    sc = torch.ones(a.cout, device=dev)
    sh = torch.zeros(a.cout, device=dev)
    d = ConvDesc()
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = a.batch, a.hw, a.hw, a.cin, a.hw, a.hw, a.cout
    d.KH = d.KW = 3
    d.stride, d.pad = 1, 1
    d.x_cstride, d.y_cstride = a.cin, a.cout
    d.act1 = 1
    d.tile = a.tile
    res = torch.randn_like(y.float()).to(tdt) if a.res else None
    res_p = res.data_ptr() if a.res else None
    if a.res:
        d.r_cstride = a.cout
    flush = torch.empty(a.flush_mb << 20, dtype=torch.uint8, device=dev) if a.flush_mb else None
    for i in range(20):
        rc = lib.yv4_conv_bn_act_fwd_h16(C.byref(d), code, code, x.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                         None, None, res_p, y.data_ptr(), stream)
        assert rc == 0, rc
    if flush is not None:
        flush.fill_(7)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lib.yv4_conv_bn_act_fwd_h16(C.byref(d), code, code, x.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, None,
                                res_p, y.data_ptr(), stream)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    n = 256 * 64
    out = (C.c_ulonglong * n)()
    assert raw.yv4_debug_w3_stamps(out, n) == 0
    st = np.frombuffer(out, dtype=np.uint64).reshape(256, 8, 8).astype(np.float64)
    live = st[:, 0, :].sum(-1) > 0
    st = st[live]
    nk = (a.cin // 64) * 9
    print(f'{a.cin}->{a.cout} @{a.hw} batch {a.batch} tile {a.tile}{" +res" if a.res else ""}{f" flush {a.flush_mb} MB" if a.flush_mb else ""}: {us:.1f} us (stamped build), {live.sum()} workgroups, {nk} K tiles per tile')
    tot = st.sum(-1)
    print(f'cycles per wave (stamped span): mean {tot.mean():.0f}  min {tot.min():.0f}  max {tot.max():.0f}')
    for role, sl in (('waves 0-3', slice(0, 4)), ('waves 4-7', slice(4, 8))):
        m = st[:, sl, :].mean((0, 1))
        line = ' '.join(f'{NAMES[i]}:{m[i]:.0f}' for i in range(8))
        print(f'{role}: {line}  total {m.sum():.0f}')
    # per K tile of the first tile a workgroup computes (most workgroups compute exactly one)
    m = st.mean((0, 1))
    print('share of the span: ' + ' '.join(f'{NAMES[i]}:{m[i] / m.sum():.3f}' for i in range(8)))


if __name__ == '__main__':
    main()
