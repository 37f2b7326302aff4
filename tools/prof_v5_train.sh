#!/bin/bash
# configs[4]: YOLOv5-L 640 bf16 train step, batch 64: throughput and per-kernel time
set -u
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_v5; rm -rf "$OUT"; mkdir -p "$OUT"
python3 tools/train_bench.py --model yolov5l --size 640 --batch 64 --steps 8 --warmup 3 --dtype bf16 2>/dev/null | python3 tools/last_json.py approx_conv_tflops peak_mem_gb
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 tools/train_bench.py --model yolov5l --size 640 --batch 64 --steps 3 --warmup 2 --dtype bf16 > "$OUT/trace.log" 2>&1 < /dev/null
f=$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then cp "$f" "$OUT/v5_train_kernel_stats.csv"; head -32 "$f" | cut -c1-150; fi
find "$OUT" -name "*kernel_trace.csv" -delete
