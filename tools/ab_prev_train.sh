# same-box A/B of the train step: lib_prev (round 5's library) against the product, 20 timed steps, three alternations
for i in 1 2 3; do
for L in prev cur; do
if [ $L = prev ]; then export YV4_LIB_PATH=$PWD/mmdet-yolov4_amd/lib_prev/libyv4_hip_prev.so YV4_LIB_ABI_ANY=1; else unset YV4_LIB_PATH YV4_LIB_ABI_ANY; fi
echo -n "$L v4l bf16 train: "; python tools/train_bench.py --batch 64 --steps 20 --warmup 5 --dtype bf16 2>/dev/null | python tools/last_json.py
done; done
