#!/bin/bash
# same-box A/B: weight gradients on a side stream (YV4_WGRAD_STREAM=1, the default) against everything on one stream
for v in 0 1 0 1; do
  echo -n "YV4_WGRAD_STREAM=$v v4l bf16 train: "; YV4_WGRAD_STREAM=$v python tools/train_bench.py --batch 64 --steps 8 --warmup 3 --dtype bf16 2>/dev/null | python tools/last_json.py
done
for v in 0 1; do
  echo -n "YV4_WGRAD_STREAM=$v v5l bf16 train: "; YV4_WGRAD_STREAM=$v python tools/train_bench.py --model yolov5l --size 640 --batch 64 --steps 8 --warmup 3 --dtype bf16 2>/dev/null | python tools/last_json.py
done
