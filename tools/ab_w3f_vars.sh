#!/bin/bash
# fp32 wide 3x3 kernel variants from lib_var (tools/build_src_variants.sh name:conv3x3_wide_f32:flags) against the product, same box:
# parity of the product first, stamps, then per layer and the headline.   VARS="name ..." (lib_var/libyv4_w3f_<name>.so)
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
L=$GRAFT_REPO_ROOT/mmdet-yolov4_amd/lib_var
VARS=${VARS:?VARS="name ..." of lib_var/libyv4_w3f_<name>.so}
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "wide or w3 or W3" 2>&1 | tail -3 || exit 1
echo "=== stamps (product source, stamped)"
YV4_LIB_PATH=$L/libyv4_w3f_stamp.so python tools/stamp_w3f.py --cin 256 --cout 256 --hw 38 2>&1 | grep -v amdgpu.ids
YV4_LIB_PATH=$L/libyv4_w3f_stamp.so python tools/stamp_w3f.py --cin 128 --cout 128 --hw 76 2>&1 | grep -v amdgpu.ids
for i in 1 2; do
for v in product $VARS; do
if [ $v = product ]; then unset YV4_LIB_PATH; else export YV4_LIB_PATH=$L/libyv4_w3f_$v.so; fi
echo "--- per layer, $v"; python tools/conv_bench.py --dtype f32 --filter k3s1 --tiles 10 --reps 5 2>/dev/null | grep "auto=w3x3"
done; done
for i in 1 2 3; do
for v in product $VARS; do
if [ $v = product ]; then unset YV4_LIB_PATH; else export YV4_LIB_PATH=$L/libyv4_w3f_$v.so; fi
echo -n "headline $v: "; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train 2>/dev/null | python tools/last_json.py roofline.frac roofline.all_convs_frac output_check
done; done
