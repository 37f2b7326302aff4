#!/bin/bash
# Border taps through out-of-range LDS reads (product) against the border select (lib_var/libyv4_<name>_sel.so, built from the
# commit before): parity first, then per layer and per configuration, same box.
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
L=$GRAFT_REPO_ROOT/mmdet-yolov4_amd/lib_var
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "wide or w3 or W3" 2>&1 | tail -3 || exit 1
for i in 1 2; do
for v in product sel; do
unset YV4_LIB_PATH; [ $v = sel ] && export YV4_LIB_PATH=$L/libyv4_w3f_sel.so
echo "--- fp32 per layer, $v"; python tools/conv_bench.py --dtype f32 --filter k3s1 --tiles 10 --reps 5 2>/dev/null | grep "auto=w3x3"
done; done
for i in 1 2 3; do
for v in product sel; do
unset YV4_LIB_PATH; [ $v = sel ] && export YV4_LIB_PATH=$L/libyv4_w3f_sel.so
echo -n "headline $v: "; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train 2>/dev/null | python tools/last_json.py roofline.frac roofline.all_convs_frac output_check
done; done
