#!/bin/bash
# per-kernel time of the post-processing kernels inside the bf16 YOLOv4-L and the fp16 YOLOv4-S (configs[3]) inference steps
set -u
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_post; rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bf16" -- python3 bench.py --dtype bf16 --steps 10 --warmup 2 --no-cpu-baseline --no-train --no-output-check > "$OUT/bf16.log" 2>&1 < /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/cfg3" -- python3 bench.py --model yolov4s --size 416 --batch 256 --dtype f16 --steps 10 --warmup 2 --no-cpu-baseline --no-train --no-output-check > "$OUT/cfg3.log" 2>&1 < /dev/null
for n in bf16 cfg3; do
  f=$(find "$OUT/$n" -name '*kernel_stats.csv' | head -1)
  echo "== $n ($f)"
  if [ -n "$f" ]; then grep -E "decode_filter|nms_images|pool5|spp_|resample|decode_reset|copyBuffer|nchw" "$f" | cut -c1-140; cp "$f" "$OUT/${n}_kernel_stats.csv"; fi
done
