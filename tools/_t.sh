timeout 600 python -m pytest tests/test_gpu_h16.py -x -q 2>&1 < /dev/null | tail -2
for i in 1 2 3; do
for v in lib_alt lib; do
echo "== $v"
YV4_LIB_PATH=$PWD/mmdet-yolov4_amd/$v/libyv4_hip.so timeout 300 python bench.py --dtype bf16 --no-cpu-baseline --steps 10 --warmup 3 2>&1 < /dev/null | tail -1 | cut -c1-150
done
done
