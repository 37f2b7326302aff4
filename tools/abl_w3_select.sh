#!/bin/bash
# 16-bit wide 3x3 kernel: the product against lib_var/libyv4_w3_abl32.so (no border handling at all: wrong results at the borders,
# timing only; tools/build_src_variants.sh w3_abl32:conv3x3_wide_h16:-DYV4_W3_ABL=32) and against the round-5 library
# and against the same kernel with the border select (lib_var/libyv4_w3_sel.so, built from the commit before), per layer; parity
# of the product first, then bf16 inference and configs[3] with the product and with the select
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
L=$GRAFT_REPO_ROOT/mmdet-yolov4_amd
timeout -k 10 900 python -m pytest tests/test_gpu_h16.py -x -q -k "wide or w3" 2>&1 | tail -3 || exit 1
for i in 1 2; do
for v in product sel abl32; do
unset YV4_LIB_PATH YV4_LIB_ABI_ANY
[ $v = abl32 ] && export YV4_LIB_PATH=$L/lib_var/libyv4_w3_abl32.so
[ $v = sel ] && export YV4_LIB_PATH=$L/lib_var/libyv4_w3_sel.so
echo "--- $v"; python tools/conv_bench.py --dtype bf16 --filter k3s1 --tiles 5 --reps 5 2>/dev/null | grep "@" | grep -v "inf"
done; done
for i in 1 2; do
for v in product sel; do
unset YV4_LIB_PATH
[ $v = sel ] && export YV4_LIB_PATH=$L/lib_var/libyv4_w3_sel.so
echo -n "bf16 inference $v: "; python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-train 2>/dev/null | python tools/last_json.py roofline.frac output_check
echo -n "configs[3] $v: "; python bench.py --model yolov4s --size 416 --batch 256 --dtype f16 --steps 20 --warmup 5 --no-cpu-baseline --no-train 2>/dev/null | python tools/last_json.py roofline.frac output_check
done; done
