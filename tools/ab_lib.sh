# same-box A/B of two builds of the library: lib_prev (the tree before a change, built by hand) against lib_alt (make measure)
for i in 1 2; do
for L in lib_prev lib_alt; do
export YV4_LIB_PATH=$PWD/mmdet-yolov4_amd/$L/libyv4_hip_measure.so
echo -n "$L v4l bf16 inference: "; python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-output-check 2>/dev/null | python tools/last_json.py
echo -n "$L v4l fp32 inference: "; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-output-check 2>/dev/null | python tools/last_json.py roofline.frac roofline.all_convs_frac
echo -n "$L cfg3: "; python bench.py --model yolov4s --size 416 --batch 256 --dtype f16 --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-output-check 2>/dev/null | python tools/last_json.py
echo -n "$L v4l bf16 train: "; python tools/train_bench.py --batch 64 --steps 8 --warmup 3 --dtype bf16 2>/dev/null | python tools/last_json.py
echo -n "$L v5l bf16 train: "; python tools/train_bench.py --model yolov5l --size 640 --batch 64 --steps 8 --warmup 3 --dtype bf16 2>/dev/null | python tools/last_json.py
done; done
