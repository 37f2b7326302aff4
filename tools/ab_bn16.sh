#!/bin/bash
# A/B of the pipelined 16-bit BatchNorm passes (train.hip bn16_*) against the general kernels, same box, measure build:
# tools/bn_bench.py --kernels over YOLOv4-L's activation shapes at batch 64.
source "$(dirname "$0")/_measure_lib.sh"
for cfg in "0 4" "1 4" "1 8" "0 4" "1 4"; do
  set -- $cfg
  echo "== YV4_BN16=$1 YV4_BN16_V=$2"
  YV4_BN16=$1 YV4_BN16_V=$2 python tools/bn_bench.py --kernels --dtype bf16 2>&1 | grep -v amdgpu.ids
done
