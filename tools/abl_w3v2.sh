#!/bin/bash
# compile-time ablations of conv_wgrad3x3_v2_h16_kernel (measure build, YV4_W3V2_ABL: 1 no DMA in the loop, 2 no workgroup
# barrier, 4 no MFMAs, 8 no fragment reads; sums of bits).  The ablated instantiations carry no extra branches.
source "$(dirname "$0")/_measure_lib.sh"
for ab in 0 1 2 3 4 8 12 5 13 15 0; do
  echo "== YV4_W3V2_ABL=$ab"
  YV4_W3V2_ABL=$ab python tools/wgrad_bench.py --filter k3s1 2>&1 | grep -E '128->128|256->256|512->512'
done
