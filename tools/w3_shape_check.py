#!/usr/bin/env python
"""Bit-identity of forced wide 3x3 tile shapes (YV4_HTILE_W3x3_SHAPE(i) = 13 + 8 i) against the 128 x 64 tile on a few layers."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from test_gpu_h16 import _h16_conv  # noqa: E402

dev = torch.device('cuda:0')
ok = True
for (N, H, Cin, Cout) in [(3, 52, 64, 64), (2, 31, 64, 64), (5, 19, 128, 64), (2, 40, 64, 32 + 32), (9, 20, 256, 64)]:
    for dtype in (torch.bfloat16, torch.float16):
        for res in (False, True):
            ref = _h16_conv(dev, dtype, N, H, H, Cin, Cout, 3, 1, 1, act=1, tile=2, raw=True, residual=res)
            for t in [int(x) for x in sys.argv[1:]]:
                out = _h16_conv(dev, dtype, N, H, H, Cin, Cout, 3, 1, 1, act=1, tile=t, raw=True, residual=res)
                same = bool(torch.equal(ref.view(torch.int16), out.view(torch.int16)))
                ok &= same
                print(N, H, Cin, Cout, dtype, 'res' if res else '', 'tile', t, 'bit-identical' if same else 'DIFFERENT')
print('all identical' if ok else 'MISMATCH')
sys.exit(0 if ok else 1)
