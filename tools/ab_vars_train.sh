#!/bin/bash
# train step with library variants (lib_var/libyv4_<name>.so) against the product, same box, alternating.  VARS="name ..."
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
L=$GRAFT_REPO_ROOT/mmdet-yolov4_amd/lib_var
for i in 1 2 3; do
for v in product ${VARS:?VARS="name ..."}; do
unset YV4_LIB_PATH; [ $v != product ] && export YV4_LIB_PATH=$L/libyv4_$v.so
echo -n "$v bf16 train: "; python tools/train_bench.py --batch 64 --steps 20 --warmup 5 --dtype bf16 2>/dev/null | python tools/last_json.py
done; done
