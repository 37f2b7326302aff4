#!/bin/bash
# A/B of the few-channel 3x3 weight-gradient kernel's second form (conv_wgrad_fc_v2_h16_kernel) against the first, measure
# build (switch YV4_WFC_V2), same box; then bit-identity of dW over the kernel's geometries (tools/wfc_bitwise.py).
source "$(dirname "$0")/_measure_lib.sh"
for v in 0 1 0 1; do
  echo "== YV4_WFC_V2=$v"
  YV4_WFC_V2=$v python tools/wgrad_bench.py --det --filter k3s1 2>&1 | grep -E '3->32|32->64|64->64'
done
for v in 0 1; do YV4_WFC_V2=$v python tools/wfc_bitwise.py gpurun_out/wfc_dw_$v.pt; done
python - <<'PY'
import torch
a, b = torch.load('gpurun_out/wfc_dw_0.pt'), torch.load('gpurun_out/wfc_dw_1.pt')
print('bit-identical dW, first vs second form:', [bool(torch.equal(x, y)) for x, y in zip(a, b)])
PY
rm -f gpurun_out/wfc_dw_0.pt gpurun_out/wfc_dw_1.pt
