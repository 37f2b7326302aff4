#!/bin/bash
# same-box A/B: the loss aggregated from the fused head's (L, 3) matrix (default) against the reference's per-tensor sums
for v in 0 1 0 1; do
  echo -n "YV4_LOSS_MATRIX=$v v4l bf16 train: "; YV4_LOSS_MATRIX=$v python tools/train_bench.py --batch 64 --steps 8 --warmup 3 --dtype bf16 2>/dev/null | python tools/last_json.py
done
