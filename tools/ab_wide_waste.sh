# the general wide kernel's fill rule (measurement build)
source "$(dirname "$0")/_measure_lib.sh"
for w in 25 35 50 25 35 50; do
echo -n "v5l bf16 inference WIDE_MAXWASTE=$w: "; YV4_WIDE_MAXWASTE=$w python bench.py --model yolov5l --size 640 --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-output-check 2>/dev/null | python tools/last_json.py
echo -n "v5l train WIDE_MAXWASTE=$w: "; YV4_WIDE_MAXWASTE=$w python tools/train_bench.py --model yolov5l --size 640 --batch 64 --steps 8 --warmup 3 --dtype bf16 2>/dev/null | python tools/last_json.py
echo -n "cfg3 WIDE_MAXWASTE=$w: "; YV4_WIDE_MAXWASTE=$w python bench.py --model yolov4s --size 416 --batch 256 --dtype f16 --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-output-check 2>/dev/null | python tools/last_json.py
echo -n "v4l bf16 inference WIDE_MAXWASTE=$w: "; YV4_WIDE_MAXWASTE=$w python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-output-check 2>/dev/null | python tools/last_json.py
echo -n "v4l train WIDE_MAXWASTE=$w: "; YV4_WIDE_MAXWASTE=$w python tools/train_bench.py --batch 64 --steps 8 --warmup 3 --dtype bf16 2>/dev/null | python tools/last_json.py
done
