# upstream MMDetection's only published numbers (configs/yolo/README.md:22-24): YOLOv3-DarkNet53 batch 1 on one V100
for sz in 320 416 608; do
for g in "" "--graph"; do
echo -n "yolov3 $sz batch 1 fp32 $g: "; python bench.py --model yolov3 --size $sz --batch 1 $g --steps 100 --warmup 10 --no-cpu-baseline --no-train --no-output-check 2>/dev/null | python tools/last_json.py vs_baseline
done; done
echo -n "yolov3 608 batch 32 fp32: "; python bench.py --model yolov3 --size 608 --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-output-check 2>/dev/null | python tools/last_json.py
echo -n "yolov3 608 batch 32 bf16: "; python bench.py --model yolov3 --size 608 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-output-check 2>/dev/null | python tools/last_json.py
