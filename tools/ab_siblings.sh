# sibling 1x1 convs as one launch (YV4_FUSE_SIBLINGS=0 keeps them apart): product library, plan-level switch
for i in 1 2; do
for f in 0 1; do
export YV4_FUSE_SIBLINGS=$f
echo -n "FUSE=$f v4l bf16 inference: "; python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-train 2>/dev/null | python tools/last_json.py output_check
echo -n "FUSE=$f v4l fp32 inference: "; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train 2>/dev/null | python tools/last_json.py output_check
echo -n "FUSE=$f cfg3: "; python bench.py --model yolov4s --size 416 --batch 256 --dtype f16 --steps 20 --warmup 5 --no-cpu-baseline --no-train 2>/dev/null | python tools/last_json.py output_check
echo -n "FUSE=$f v5l bf16 inference: "; python bench.py --model yolov5l --size 640 --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-output-check 2>/dev/null | python tools/last_json.py
done; done
