#!/bin/bash
# in-step A/B of the pipelined 16-bit BatchNorm passes: bf16 train step, YOLOv4-L 608 batch 64, measure build, same box
source "$(dirname "$0")/_measure_lib.sh"
for v in 0 1 0 1; do
  echo -n "YV4_BN16=$v v4l bf16 train: "; YV4_BN16=$v python tools/train_bench.py --batch 64 --steps 8 --warmup 3 --dtype bf16 2>/dev/null | python tools/last_json.py
done
