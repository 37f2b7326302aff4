#!/bin/bash
# A/B of the 3x3 weight-gradient kernel's second form (conv_wgrad3x3_v2_h16_kernel) against the first, measure build, same
# box: per-layer times, then bit-identity of the two kernels' dW on a few layers (tools/w3g_bitwise.py).
source "$(dirname "$0")/_measure_lib.sh"
for v in 0 1 0 1; do
  echo "== YV4_W3V2=$v"
  YV4_W3V2=$v python tools/wgrad_bench.py --det --filter k3s1 2>&1 | grep -v amdgpu.ids
done
echo "== network, YV4_W3V2=0 / 1"
YV4_W3V2=0 python tools/wgrad_bench.py --det 2>&1 | tail -1
YV4_W3V2=1 python tools/wgrad_bench.py --det 2>&1 | tail -1
for v in 0 1; do YV4_W3V2=$v python tools/w3g_bitwise.py gpurun_out/w3g_dw_$v.pt; done
python - <<'PY'
import torch
a, b = torch.load('gpurun_out/w3g_dw_0.pt'), torch.load('gpurun_out/w3g_dw_1.pt')
print('bit-identical dW, first vs second form:', [bool(torch.equal(x, y)) for x, y in zip(a, b)])
PY
rm -f gpurun_out/w3g_dw_0.pt gpurun_out/w3g_dw_1.pt
