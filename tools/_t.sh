B="timeout 300 python bench.py --no-cpu-baseline --steps 5 --warmup 2"
for args in "--batch 1" "--batch 8 --size 416" "--batch 64" "--model yolov5l --size 640 --batch 16" "--model yolov5l --size 640 --batch 16 --dtype bf16" "--model yolov4s --size 416 --batch 64 --dtype f16" "--model yolov3 --batch 8 --dtype bf16" "--batch 5 --size 512 --no-autotune"; do
echo "== $args"; $B $args 2>&1 < /dev/null | tail -1 | cut -c1-130
done
T="timeout 300 python tools/train_bench.py --steps 3 --warmup 2"
for args in "--batch 4 --size 416" "--model yolov5l --size 640 --batch 8 --dtype bf16" "--model yolov3 --batch 4 --dtype f16" "--model yolov4s --size 416 --batch 16"; do
echo "== train $args"; $T $args 2>&1 < /dev/null | tail -1 | cut -c1-130
done
