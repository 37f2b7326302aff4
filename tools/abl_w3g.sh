#!/bin/bash
# additive ablation of the 3x3 weight-gradient kernel (conv_wgrad3x3_h16_kernel), measure build:
# YV4_W3_ABLATE bits: 1 no MFMAs, 2 no fragment reads, 4 no DMA after the prologue, 8 no border masks, 16 no output
source "$(dirname "$0")/_measure_lib.sh"
for ab in 0 1 2 3 4 8 16 7 15 31; do
  echo "== YV4_W3_ABLATE=$ab"
  YV4_W3_ABLATE=$ab python tools/wgrad_bench.py --det --filter k3s1 2>&1 | grep -E '128->128|256->256|512->512|256->512 k3s1'
done
