#!/usr/bin/env python
"""Micro-benchmark of yv4_conv_bn_act_fwd over the YOLOv4-L layer shapes x tile configs.
Usage (GPU box):  python tools/conv_bench.py [--batch 32] [--tiles 1,2,3,4] [--filter 3x3]
Prints one line per (shape, tile): microseconds and TFLOP/s (HIP events, median of reps)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mmdet_yolov4_amd as pkg  # noqa: E402
from mmdet_yolov4_amd._lib import ConvDesc, TILE_NAMES  # noqa: E402

# (Cin, Cout, k, stride, Hin) of YOLOv4-L @608, count
SHAPES = [
    (3, 32, 3, 1, 608, 1), (32, 64, 3, 2, 608, 1), (64, 32, 1, 1, 304, 1), (32, 64, 3, 1, 304, 1),
    (64, 128, 3, 2, 304, 1), (128, 64, 1, 1, 152, 2), (64, 64, 1, 1, 152, 3), (64, 64, 3, 1, 152, 2),
    (128, 128, 1, 1, 152, 1), (128, 256, 3, 2, 152, 1), (256, 128, 1, 1, 76, 5), (128, 128, 1, 1, 76, 12),
    (128, 128, 3, 1, 76, 10), (256, 256, 1, 1, 76, 1), (256, 512, 3, 2, 76, 1), (512, 256, 1, 1, 38, 7),
    (256, 256, 1, 1, 38, 15), (256, 256, 3, 1, 38, 12), (512, 512, 1, 1, 38, 1), (512, 1024, 3, 2, 38, 1),
    (1024, 512, 1, 1, 19, 7), (512, 512, 1, 1, 19, 9), (512, 512, 3, 1, 19, 8), (1024, 1024, 1, 1, 19, 1),
    (2048, 512, 1, 1, 19, 1), (128, 256, 3, 1, 76, 1), (256, 512, 3, 1, 38, 1), (512, 1024, 3, 1, 19, 1),
    (128, 256, 3, 2, 76, 1), (256, 512, 3, 2, 38, 1), (256, 512, 1, 1, 38, 2), (256, 255, 1, 1, 76, 1), (512, 255, 1, 1, 38, 1),
    (1024, 255, 1, 1, 19, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--tiles', default='1,2,3,4')
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--filter', default='')
    ap.add_argument('--json', default='')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'f16', 'bf16'])
    ap.add_argument('--chain', type=int, default=1, help='launches per timed region (steady-state time per launch)')
    ap.add_argument('--warm-ms', type=float, default=30.0, help='untimed launches of a tile before it is timed (ms of wall time)')
    ap.add_argument('--act', type=int, default=1, help='activation id of the epilogue (0 none, 1 Mish, 2 leaky, 3 swish)')
    ap.add_argument('--flush-mb', type=int, default=0, help='overwrite a buffer of this many MB before every timed launch (64: the '
                    'per-XCD L2s forget the layer, the memory-side cache keeps it: the state a layer meets inside a network)')
    ap.add_argument('--res', action='store_true', help='with a residual tensor added in the epilogue (the Bottleneck 3x3 layers)')
    ap.add_argument('--shape', action='append', default=[], help='cin,cout,k,stride,hin: time this shape instead of the YOLOv4-L table (repeatable)')
    ap.add_argument('--zeros', action='store_true', help='all-zero operands: the clock the chip holds on trivial data (DVFS check)')
    a = ap.parse_args()
    tiles = [int(t) for t in a.tiles.split(',')]
    dev = torch.device('cuda:0')
    lib = pkg._lib.lib()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    NAMES = pkg._lib.HTILE_NAMES if a.dtype != 'f32' else TILE_NAMES
    rows = []
    tot = {t: 0.0 for t in tiles}
    best_tot = 0.0
    shapes = [tuple(int(v) for v in sh.split(',')) + (1,) for sh in a.shape] or SHAPES
    for (cin, cout, k, s, h, cnt) in shapes:
        tag = f'{cin}->{cout} k{k}s{s} @{h}'
        if a.filter and a.filter not in tag:
            continue
        h16 = a.dtype != 'f32'
        tdt = dict(f32=torch.float32, f16=torch.float16, bf16=torch.bfloat16)[a.dtype]
        cp = (cin + 7) // 8 * 8 if h16 else (cin + 3) // 4 * 4
        pad = k // 2
        ho = (h + 2 * pad - k) // s + 1
        x = torch.randn(a.batch * h * h * cp, device=dev).to(tdt)
        w = (torch.randn(cout * k * k * cp, device=dev) * 0.05).to(tdt)
        if a.zeros:
            x.zero_()
            w.zero_()
        ycs = (cout + 7) // 8 * 8 if h16 else cout
        y = torch.empty(a.batch * ho * ho * ycs, device=dev, dtype=tdt)
        sc = torch.ones(cout, device=dev)
        sh = torch.zeros(cout, device=dev)
        d = ConvDesc()
        d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = a.batch, h, h, cp, ho, ho, cout
        d.KH = d.KW = k
        d.stride, d.pad = s, pad
        d.x_cstride, d.y_cstride = cp, (ycs if h16 else cout)
        d.act1 = a.act
        res_t = torch.randn_like(y.float()).to(tdt) if a.res else None
        res_p = res_t.data_ptr() if a.res else None
        if a.res:
            d.r_cstride = (ycs if h16 else cout)
        flops = 2.0 * a.batch * ho * ho * cout * k * k * cin
        res = {}
        def launch(code_=None):
            if h16:
                code = 1 if a.dtype == 'f16' else 2
                return lib.yv4_conv_bn_act_fwd_h16(C.byref(d), code, code, x.data_ptr(), w.data_ptr(), sc.data_ptr(),
                                                   sh.data_ptr(), None, None, res_p, y.data_ptr(), stream)
            return lib.yv4_conv_bn_act_fwd(C.byref(d), x.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, None,
                                           res_p, y.data_ptr(), stream)

        for t in tiles:
            d.tile = t
            # untimed launches for --warm-ms of wall time first: the first tile measured after the host-side set-up of a
            # layer ran 1-15 % slower than the same tile measured later (clock ramp), which biased every comparison
            # against the tile listed first
            t0 = time.time()
            while launch() == 0 and (time.time() - t0) * 1e3 < a.warm_ms:
                torch.cuda.synchronize()
            torch.cuda.synchronize()
            ts = []
            for r in range(a.reps + 1):
                if a.flush_mb:
                    if not hasattr(main, '_flush'):
                        main._flush = torch.empty(a.flush_mb << 20, dtype=torch.uint8, device=dev)
                    main._flush.fill_(r & 255)
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _c in range(a.chain):
                    if h16:
                        code = 1 if a.dtype == 'f16' else 2
                        rc = lib.yv4_conv_bn_act_fwd_h16(C.byref(d), code, code, x.data_ptr(), w.data_ptr(),
                                                         sc.data_ptr(), sh.data_ptr(), None, None, res_p, y.data_ptr(),
                                                         stream)
                    else:
                        rc = lib.yv4_conv_bn_act_fwd(C.byref(d), x.data_ptr(), w.data_ptr(), sc.data_ptr(),
                                                     sh.data_ptr(), None, None, res_p, y.data_ptr(), stream)
                e1.record()
                torch.cuda.synchronize()
                if rc != 0:
                    break
                if r:
                    ts.append(e0.elapsed_time(e1) * 1e3 / a.chain)
            ts.sort()
            res[t] = ts[len(ts) // 2] if ts else float('inf')
            tot[t] += res[t] * cnt
        d.tile = 0
        auto = (lib.yv4_conv_h16_pick_tile if h16 else lib.yv4_conv_pick_tile)(C.byref(d))
        best = min(res, key=res.get)
        best_tot += res[best] * cnt
        line = f'{tag:28s} x{cnt:2d} ' + ' '.join(
            f'{NAMES.get(t, t)}:{res[t]:7.0f}us {flops / res[t] / 1e6:6.1f}TF' for t in tiles)
        print(line + f'  best={NAMES.get(best, best)} auto={NAMES.get(auto, auto)}', flush=True)
        rows.append(dict(shape=[cin, cout, k, s, h], count=cnt, us=res, best=best, auto=auto))
    print('weighted totals (us): ' + ' '.join(f'{NAMES.get(t, t)}:{v:.0f}' for t, v in tot.items()) +
          f' best-per-shape:{best_tot:.0f}')
    if a.json:
        json.dump(rows, open(a.json, 'w'))


if __name__ == '__main__':
    main()
