"""dW of few-channel 3x3 / stride-1 layers (the domain of conv_wgrad_fc_h16_kernel: Cin 16 / 32 / 64 per pixel, Cout 32 / 64,
>= 131 072 rows) through yv4_conv_wgrad_det, saved for a bitwise comparison between two switches of the library (tools/ab_wfc.sh)."""
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
    for (n, cin, cout, h, w) in ((4, 32, 64, 192, 192), (3, 16, 32, 224, 208), (2, 64, 64, 260, 256), (9, 32, 32, 121, 123),
                                 (2, 64, 32, 300, 230), (5, 16, 64, 170, 160), (40, 32, 64, 61, 59)):
        g = torch.Generator(device='cpu').manual_seed(cin + h)
        x = torch.randn(n * h * w * cin, generator=g).to(dev).to(dt)
        dy = torch.randn(n * h * w * cout, generator=g).to(dev).to(dt)
        dw = torch.zeros(cout * 9 * cin, device=dev)
        d = ConvDesc()
        d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = n, h, w, cin, h, w, cout
        d.KH = d.KW = 3
        d.stride, d.pad = 1, 1
        d.x_cstride, d.y_cstride = cin, cout
        need = int(lib.yv4_conv_wgrad_workspace(C.byref(d), code))
        ws = torch.empty(max(need // 4, 4), device=dev)
        rc = lib.yv4_conv_wgrad_det(C.byref(d), code, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), need, stream)
        assert rc == 0, lib.yv4_last_error()
        torch.cuda.synchronize()
        out.append(dw.cpu())
torch.save(out, sys.argv[1])
