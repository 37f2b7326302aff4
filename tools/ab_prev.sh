# same-box A/B of two PRODUCT builds: mmdet-yolov4_amd/lib_prev/libyv4_hip_prev.so (the tree before a change: `git archive HEAD`
# built by hand into lib_prev/, travels with the push) against lib/libyv4_hip.so.   bash tools/ab_prev.sh [legs]
LEGS=${1:-"bf16 f32 cfg3 train"}
for i in 1 2; do
for L in prev cur; do
if [ $L = prev ]; then export YV4_LIB_PATH=$PWD/mmdet-yolov4_amd/lib_prev/libyv4_hip_prev.so YV4_LIB_ABI_ANY=1; else unset YV4_LIB_PATH YV4_LIB_ABI_ANY; fi
for leg in $LEGS; do
case $leg in
bf16) echo -n "$L v4l bf16 inference: "; python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-output-check 2>/dev/null | python tools/last_json.py;;
f32) echo -n "$L v4l fp32 inference: "; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-output-check 2>/dev/null | python tools/last_json.py roofline.frac roofline.all_convs_frac;;
cfg3) echo -n "$L cfg3: "; python bench.py --model yolov4s --size 416 --batch 256 --dtype f16 --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-output-check 2>/dev/null | python tools/last_json.py;;
train) echo -n "$L v4l bf16 train: "; python tools/train_bench.py --batch 64 --steps 8 --warmup 3 --dtype bf16 2>/dev/null | python tools/last_json.py;;
esac
done; done; done
