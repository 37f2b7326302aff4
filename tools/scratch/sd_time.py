import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mmdet_yolov4_amd as pkg
from mmdet_yolov4_amd import _lib as L
dev = torch.device('cuda:0')
N, H, W, C1, C2 = (int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (32, 608, 608, 32, 64)))
dt = torch.bfloat16
x = torch.rand(N, 3, H, W, device=dev)
w1 = torch.randn(C1, 36, device=dev) * 0.2
w2 = (torch.randn(C2, 9 * C1, device=dev) * 0.06).to(dt)
s1, t1, s2, t2 = torch.ones(C1, device=dev), torch.zeros(C1, device=dev), torch.ones(C2, device=dev), torch.zeros(C2, device=dev)
y = torch.empty(N, H // 2, W // 2, C2, dtype=dt, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run():
    L.check(L.lib().yv4_stem_down_fwd_h16(2, x.data_ptr(), N, H, W, w1.data_ptr(), s1.data_ptr(), t1.data_ptr(), C1, 1, 0.0,
                                          w2.data_ptr(), s2.data_ptr(), t2.data_ptr(), C2, 1, 0.0, y.data_ptr(), C2, 0, st), 'sd')
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print(f'ablate={os.environ.get("YV4_SD_ABLATE","0"):>3s}  {e0.elapsed_time(e1) * 100:.1f} us')
