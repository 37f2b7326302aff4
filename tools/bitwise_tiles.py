"""Do the generic fp32 LDS-DMA tiles (ids 5, 6, 7) give the same bits on one layer?  And does this library give the same
bits as another build (YV4_LIB_PATH) -- run twice and compare the printed checksums.
    python tools/bitwise_tiles.py"""
import ctypes as C
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mmdet_yolov4_amd as pkg  # noqa: E402
from mmdet_yolov4_amd._lib import ConvDesc  # noqa: E402

dev = torch.device('cuda:0')
lib = pkg._lib.lib()
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (n, cin, cout, k, s, h) in [(8, 32, 64, 3, 1, 304), (8, 32, 64, 3, 2, 608), (8, 64, 64, 3, 1, 152), (8, 256, 256, 3, 1, 38),
                                (8, 128, 128, 1, 1, 76), (8, 512, 256, 1, 1, 38)]:
    g = torch.Generator().manual_seed(1)
    pad = k // 2
    ho = (h + 2 * pad - k) // s + 1
    x = torch.randn(n * h * h * cin, generator=g).to(dev)
    w = (torch.randn(cout * k * k * cin, generator=g) * 0.05).to(dev)
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    outs = {}
    for t in (5, 6, 7):
        y = torch.zeros(n * ho * ho * cout, device=dev)
        d = ConvDesc()
        d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = n, h, h, cin, ho, ho, cout
        d.KH = d.KW = k
        d.stride, d.pad = s, pad
        d.x_cstride, d.y_cstride = cin, cout
        d.act1 = 1
        d.tile = t
        rc = lib.yv4_conv_bn_act_fwd(C.byref(d), x.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, None, None,
                                     y.data_ptr(), stream)
        torch.cuda.synchronize()
        assert rc == 0, rc
        outs[t] = y
    same = {t: bool(torch.equal(outs[5], outs[t])) for t in (6, 7)}
    md5 = hashlib.md5(outs[6].cpu().numpy().tobytes()).hexdigest()[:12]
    print(f'{cin}->{cout} k{k}s{s} @{h}: tile 6 == tile 5: {same[6]}, tile 7 == tile 5: {same[7]}, md5(tile 6) {md5}', flush=True)
