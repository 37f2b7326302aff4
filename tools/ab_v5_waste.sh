# the wide 3x3 kernel's rule against the ping-pong kernel (measurement build: YV4_W3_VSPP=0 = the fill rule alone)
source "$(dirname "$0")/_measure_lib.sh"
for w in 0 1 0 1; do
echo -n "v5l train VSPP=$w: "; YV4_W3_VSPP=$w python tools/train_bench.py --model yolov5l --size 640 --batch 64 --steps 8 --warmup 3 --dtype bf16 2>/dev/null | python tools/last_json.py
echo -n "v5l bf16 inference VSPP=$w: "; YV4_W3_VSPP=$w python bench.py --model yolov5l --size 640 --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-output-check 2>/dev/null | python tools/last_json.py
echo -n "cfg3 VSPP=$w: "; YV4_W3_VSPP=$w python bench.py --model yolov4s --size 416 --batch 256 --dtype f16 --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-output-check 2>/dev/null | python tools/last_json.py
echo -n "v4l bf16 inference VSPP=$w: "; YV4_W3_VSPP=$w python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-output-check 2>/dev/null | python tools/last_json.py
echo -n "v4l train VSPP=$w: "; YV4_W3_VSPP=$w python tools/train_bench.py --batch 64 --steps 8 --warmup 3 --dtype bf16 2>/dev/null | python tools/last_json.py
done
