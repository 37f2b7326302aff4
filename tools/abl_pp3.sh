#!/bin/bash
# Ablation / phase-timing runs of the persistent 3x3 kernel (measurement build; see profiles/r03_pp3_ablation.md).
# usage (GPU box): bash tools/abl_pp3.sh "<bits> <bits> ..."  ["<layer filter>" ...]
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
M=mmdet-yolov4_amd/lib_alt/libyv4_hip_measure.so
BITS="${1:-0 1 2 4 8 16 3 6 7}"
shift || true
if [ $# -eq 0 ]; then set -- "256->256 k3s1" "128->128 k3s1 @76" "512->512 k3s1"; fi
for f in "$@"; do
  for ab in $BITS; do
    echo "== $f ablate $ab"
    YV4_LIB_PATH=$M YV4_H16_ABLATE=$ab timeout -k 5 100 python tools/conv_bench.py --dtype bf16 --tiles 4 --filter "$f" --chain 4 --reps 3 2>&1 | grep -E "pp3 wg 0 wave [04]|us " | tail -3
  done
done
