#!/bin/bash
# 16-bit wide 3x3 kernel: variants from lib_var (tools/build_src_variants.sh name:conv3x3_wide_h16:flags) against the product, per
# layer and in the network, same box.   VARS="name ..." (lib_var/libyv4_<name>.so)
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
L=$GRAFT_REPO_ROOT/mmdet-yolov4_amd/lib_var
VARS=${VARS:?VARS="name ..."}
for i in 1 2; do
for v in product $VARS; do
unset YV4_LIB_PATH; [ $v != product ] && export YV4_LIB_PATH=$L/libyv4_$v.so
echo "--- $v"; python tools/conv_bench.py --dtype bf16 --filter k3s1 --tiles 5 --reps 5 2>/dev/null | grep "@" | grep -v "inf"
done; done
for i in 1 2; do
for v in product $VARS; do
unset YV4_LIB_PATH; [ $v != product ] && export YV4_LIB_PATH=$L/libyv4_$v.so
echo -n "bf16 inference $v: "; python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-train 2>/dev/null | python tools/last_json.py roofline.frac output_check
done; done
