#!/bin/bash
# 16-bit weight-stationary 1x1 kernel: strips that do not drain their loads in front of their stores (product) against the drain
# (lib_var/libyv4_ws_drain.so: tools/build_src_variants.sh ws_drain:conv1x1_ws_h16:-DYV4_WS_NODRAIN=0); parity first, then per
# layer (YOLOv4-L shapes, tile 6) and in the networks, same box
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
L=$GRAFT_REPO_ROOT/mmdet-yolov4_amd/lib_var
timeout -k 10 900 python -m pytest tests/test_gpu_h16.py -x -q -k "ws or 1x1 or pointwise" 2>&1 | tail -3 || exit 1
for i in 1 2; do
for v in product ws_drain; do
unset YV4_LIB_PATH; [ $v != product ] && export YV4_LIB_PATH=$L/libyv4_$v.so
echo "--- $v"; python tools/conv_bench.py --dtype bf16 --filter k1s1 --tiles 6 --reps 5 2>/dev/null | grep "auto=h16_ws"
done; done
for i in 1 2; do
for v in product ws_drain; do
unset YV4_LIB_PATH; [ $v != product ] && export YV4_LIB_PATH=$L/libyv4_$v.so
echo -n "$v bf16 inference: "; python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-train 2>/dev/null | python tools/last_json.py roofline.frac output_check
echo -n "$v cfg3: "; python bench.py --model yolov4s --size 416 --batch 256 --dtype f16 --steps 20 --warmup 5 --no-cpu-baseline --no-train 2>/dev/null | python tools/last_json.py roofline.frac output_check
done; done
