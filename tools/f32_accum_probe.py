#!/usr/bin/env python
"""is the fp32 path's distance from float64 (1.3-1.7 x the CPU oracle's own) the price of ONE
accumulation chain over K = 4 608?

One layer, 512 -> 512 3x3 @19 (K = 4 608), random data, no activation, against the float64 convolution:
  * the HIP kernel as the plan runs it (wide 3x3 tile: one accumulator per output, K walked in one chain; and the
    32x32x2 LDS-DMA tile: two alternating accumulator sets),
  * the same layer as TWO launches over the two halves of the input channels, added in fp32 -- a two-chain (pairwise)
    K split, what a two-accumulator kernel would compute (up to which products land in which chain),
  * four launches over channel quarters,
  * the CPU oracle's arithmetic (torch conv2d, fp32, CPU).
Prints mean / max of |y - y64| / (1 + |y64|) and the ratio to the CPU oracle's."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mmdet_yolov4_amd as pkg  # noqa: E402
from mmdet_yolov4_amd._lib import ConvDesc  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    lib = pkg._lib.lib()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    torch.manual_seed(0)
    N, H, Cin, Cout = 8, 19, 512, 512
    x = torch.randn(N, Cin, H, H)
    w = torch.randn(Cout, Cin, 3, 3) * (2.0 / (9 * Cin)) ** 0.5
    y64 = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    ycpu = torch.nn.functional.conv2d(x, w, padding=1)
    xn = x.permute(0, 2, 3, 1).contiguous().to(dev)                      # NHWC
    ones, zeros = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)

    def launch(c0, c1, tile):
        wp = w[:, c0:c1].permute(0, 2, 3, 1).contiguous().to(dev)        # [Cout][KH][KW][Cin part]
        y = torch.empty(N, H, H, Cout, device=dev)
        d = ConvDesc()
        d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = N, H, H, c1 - c0, H, H, Cout
        d.KH = d.KW = 3
        d.stride, d.pad = 1, 1
        d.x_cstride, d.x_coff, d.y_cstride = Cin, c0, Cout
        d.tile = tile
        rc = lib.yv4_conv_bn_act_fwd(C.byref(d), xn.data_ptr(), wp.data_ptr(), ones.data_ptr(), zeros.data_ptr(), None, None,
                                     None, y.data_ptr(), stream)
        assert rc == 0, lib.yv4_last_error()
        torch.cuda.synchronize()
        return y.permute(0, 3, 1, 2).cpu()

    def err(y):
        e = (y.double() - y64).abs() / (1 + y64.abs())
        return float(e.mean()), float(e.max())

    rows = [('CPU oracle arithmetic (torch conv2d fp32)', err(ycpu))]
    for tile, name in ((10, 'wide 3x3 tile (16x16x4 MFMA, one accumulator)'), (6, 'LDS-DMA 128x64 tile (32x32x2, two accumulator sets)')):
        rows.append((f'HIP {name}, one launch', err(launch(0, Cin, tile))))
        rows.append((f'HIP {name}, 2 launches over channel halves, summed', err(launch(0, 256, tile) + launch(256, 512, tile))))
        q = sum(launch(128 * i, 128 * (i + 1), tile) for i in range(4))
        rows.append((f'HIP {name}, 4 launches over channel quarters, summed', err(q)))
    base = rows[0][1]
    print(f'512 -> 512 3x3 @19, K = 4608, batch {N}: |y - y64| / (1 + |y64|)')
    for name, (m, x_) in rows:
        print(f'  {name:86s} mean {m:.3e} ({m / base[0]:.2f} x oracle)  max {x_:.3e} ({x_ / base[1]:.2f} x)')


if __name__ == '__main__':
    main()
