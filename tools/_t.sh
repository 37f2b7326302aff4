timeout 900 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_syncbn.py tests/test_gpu_train_parity.py tests/test_gpu_ddp.py -x -q 2>&1 < /dev/null | tail -2
for i in 1 2 3; do
for v in lib_alt lib; do
echo "== $v"
YV4_LIB_PATH=$PWD/mmdet-yolov4_amd/$v/libyv4_hip.so timeout 300 python tools/train_bench.py --dtype bf16 --batch 64 --steps 8 --warmup 3 2>&1 < /dev/null | tail -1 | cut -c1-120
done
done
