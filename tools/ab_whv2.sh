#!/bin/bash
# A/B of the generic 16-bit weight-gradient kernel's second form (conv_wgrad_v2_h16_kernel) against the first, measure build,
# same box; then bit-identity of dW on 1x1, stride-2 and odd shapes (tools/whv2_bitwise.py).
source "$(dirname "$0")/_measure_lib.sh"
for v in 0 1 0 1; do
  echo "== YV4_WGRAD_V2=$v"
  YV4_WGRAD_V2=$v python tools/wgrad_bench.py --det 2>&1 | grep -E 'k1s1|k3s2|weighted'
done
for v in 0 1; do YV4_WGRAD_V2=$v python tools/whv2_bitwise.py gpurun_out/whv2_dw_$v.pt; done
python - <<'PY'
import torch
a, b = torch.load('gpurun_out/whv2_dw_0.pt'), torch.load('gpurun_out/whv2_dw_1.pt')
print('bit-identical dW, first vs second form:', [bool(torch.equal(x, y)) for x, y in zip(a, b)])
PY
rm -f gpurun_out/whv2_dw_0.pt gpurun_out/whv2_dw_1.pt
