timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_h16.py tests/test_gpu_train_ops.py tests/test_gpu_fullsize.py -x -q 2>&1 < /dev/null | tail -3
for i in 1 2; do
for v in lib_alt lib; do
echo "== $v"
YV4_LIB_PATH=$PWD/mmdet-yolov4_amd/$v/libyv4_hip.so timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>&1 < /dev/null | tail -1 | cut -c1-150
YV4_LIB_PATH=$PWD/mmdet-yolov4_amd/$v/libyv4_hip.so timeout 300 python bench.py --dtype bf16 --no-cpu-baseline --steps 10 --warmup 3 2>&1 < /dev/null | tail -1 | cut -c1-150
done
done
for v in lib_alt lib; do
YV4_LIB_PATH=$PWD/mmdet-yolov4_amd/$v/libyv4_hip.so timeout 300 python tools/train_bench.py --dtype bf16 --batch 64 --steps 8 --warmup 3 2>&1 < /dev/null | tail -1 | cut -c1-120
done
