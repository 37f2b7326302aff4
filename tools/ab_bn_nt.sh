#!/bin/bash
# non-temporal row loads (1) / stores (2) / both (3) in the BatchNorm passes against the product library: build-time variants
#   make OUT=../lib_ntV LIBNAME=libyv4_ntV.so EXTRA=-DYV4_BN_NT=V   (V = 1, 2, 3; built by hand, they travel with the push)
# the passes alone (tools/bn_bench.py --kernels) and the bf16 train step, same box
cd "$(dirname "$0")/.."
for i in 1 2; do
for v in 0 1 2 3; do
  if [ $v = 0 ]; then unset YV4_LIB_PATH; else export YV4_LIB_PATH=$PWD/mmdet-yolov4_amd/lib_nt$v/libyv4_nt$v.so; fi
  echo "== YV4_BN_NT=$v"
  python tools/bn_bench.py --kernels --batch 64 --dtype bf16 2>&1 | grep 'weighted'
  echo -n "   v4l bf16 train: "; python tools/train_bench.py --batch 64 --steps 8 --warmup 3 --dtype bf16 2>/dev/null | python tools/last_json.py
done; done
