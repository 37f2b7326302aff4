#!/bin/bash
# a whole-library variant (lib_var/libyv4_<name>.so) against the product in the networks, same box.  VAR=name [LEGS="bf16 cfg3 f32 train"]
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
V=$GRAFT_REPO_ROOT/mmdet-yolov4_amd/lib_var/libyv4_${VAR:?VAR=name}.so
LEGS=${LEGS:-"bf16 cfg3 train"}
for i in 1 2; do
for v in product var; do
unset YV4_LIB_PATH; [ $v = var ] && export YV4_LIB_PATH=$V
for leg in $LEGS; do
case $leg in
bf16) echo -n "$v bf16 inference: "; python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-train 2>/dev/null | python tools/last_json.py roofline.frac output_check;;
f32) echo -n "$v fp32 inference: "; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train 2>/dev/null | python tools/last_json.py roofline.frac output_check;;
cfg3) echo -n "$v cfg3: "; python bench.py --model yolov4s --size 416 --batch 256 --dtype f16 --steps 20 --warmup 5 --no-cpu-baseline --no-train 2>/dev/null | python tools/last_json.py roofline.frac output_check;;
train) echo -n "$v bf16 train: "; python tools/train_bench.py --batch 64 --steps 20 --warmup 5 --dtype bf16 2>/dev/null | python tools/last_json.py;;
esac
done; done; done
