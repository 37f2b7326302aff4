#!/bin/bash
# bench.py with the detections' copy on the step's stream (--copy-stream 0) and on its own stream (1, the default), same box
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
for S in 0 1; do
echo -n "copy-stream $S fp32: "; python bench.py --copy-stream $S --steps 20 --warmup 5 --no-cpu-baseline --no-train 2>/dev/null | python tools/last_json.py roofline.frac output_check
echo -n "copy-stream $S bf16: "; python bench.py --copy-stream $S --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-train 2>/dev/null | python tools/last_json.py roofline.frac output_check
echo -n "copy-stream $S cfg3: "; python bench.py --copy-stream $S --model yolov4s --size 416 --batch 256 --dtype f16 --steps 20 --warmup 5 --no-cpu-baseline --no-train 2>/dev/null | python tools/last_json.py roofline.frac output_check
done; done
echo -n "copy-stream 1, streams 2, cfg3: "; python bench.py --streams 2 --model yolov4s --size 416 --batch 256 --dtype f16 --steps 20 --warmup 5 --no-cpu-baseline --no-train 2>/dev/null | python tools/last_json.py roofline.frac output_check
echo -n "copy-stream 1, streams 2, fp32: "; python bench.py --streams 2 --steps 20 --warmup 5 --no-cpu-baseline --no-train 2>/dev/null | python tools/last_json.py roofline.frac output_check
