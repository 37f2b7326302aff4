#!/bin/bash
# compile-time ablation of the fp32 wide 3x3 kernel's K loop (lib_var/libyv4_w3f_abl<bits>.so: tools/build_src_variants.sh
# w3f_abl<bits>:conv3x3_wide_f32:-DYV4_W3F_ABL=<bits>): 1 no image pieces, 2 no weight pieces, 4 no border select, 8 no counted
# wait, 16 no barrier.  Per layer, us.
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
L=$GRAFT_REPO_ROOT/mmdet-yolov4_amd/lib_var
for i in 1 2; do
for v in product abl1 abl2 abl3 abl4 abl11 abl27 abl31; do
if [ $v = product ]; then unset YV4_LIB_PATH; else export YV4_LIB_PATH=$L/libyv4_w3f_$v.so; fi
echo "--- $v"; python tools/conv_bench.py --dtype f32 --filter k3s1 --tiles 10 --reps 5 2>/dev/null | grep "auto=w3x3"
done; done
