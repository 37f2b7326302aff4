#!/bin/bash
# conv_wgrad3x3_v2_h16_kernel with and without the border masks of its image reads (measure build, YV4_W3V2_ABL=16)
source "$(dirname "$0")/_measure_lib.sh"
for i in 1 2; do
for ab in 0 16; do
  echo "== YV4_W3V2_ABL=$ab"
  YV4_W3V2_ABL=$ab python tools/wgrad_bench.py --filter k3s1 2>&1 | grep -E '128->128|256->256|512->512|->256 |->512 |->1024'
done; done
