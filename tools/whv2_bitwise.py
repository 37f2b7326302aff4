"""dW of 1x1, stride-2 3x3 and odd-shaped layers through yv4_conv_wgrad_det (bf16 and fp16), saved for a bitwise comparison
between two builds / switches of the library (tools/ab_whv2.sh)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mmdet_yolov4_amd as pkg  # noqa: E402
from mmdet_yolov4_amd._lib import ConvDesc  # noqa: E402

dev = torch.device('cuda:0')
lib = pkg._lib.lib()
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
out = []
for code, dt in ((2, torch.bfloat16), (1, torch.float16)):
    # n, cin, cout, h, w, k, s
    for (n, cin, cout, h, w, k, s) in ((8, 128, 128, 76, 76, 1, 1), (4, 256, 192, 19, 21, 1, 1), (16, 64, 128, 152, 152, 3, 2),
                                       (3, 128, 256, 37, 41, 3, 2), (2, 264, 136, 17, 23, 1, 1), (2, 72, 40, 33, 29, 3, 2),
                                       (5, 32, 64, 64, 48, 3, 2), (2, 24, 16, 40, 40, 5, 1)):
        pad = k // 2
        ho, wo = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
        g = torch.Generator(device='cpu').manual_seed(cin + h + k)
        x = torch.randn(n * h * w * cin, generator=g).to(dev).to(dt)
        dy = torch.randn(n * ho * wo * cout, generator=g).to(dev).to(dt)
        dw = torch.zeros(cout * k * k * cin, device=dev)
        d = ConvDesc()
        d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = n, h, w, cin, ho, wo, cout
        d.KH = d.KW = k
        d.stride, d.pad = s, pad
        d.x_cstride, d.y_cstride = cin, cout
        need = int(lib.yv4_conv_wgrad_workspace(C.byref(d), code))
        ws = torch.empty(max(need // 4, 4), device=dev)
        rc = lib.yv4_conv_wgrad_det(C.byref(d), code, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), need, stream)
        assert rc == 0, lib.yv4_last_error()
        torch.cuda.synchronize()
        out.append(dw.cpu())
torch.save(out, sys.argv[1])
