for i in 1 2 3; do
for v in lib_alt lib; do
echo "== $v"
YV4_LIB_PATH=$PWD/mmdet-yolov4_amd/$v/libyv4_hip.so timeout 300 python tools/train_bench.py --dtype bf16 --batch 64 --steps 10 --warmup 3 2>&1 < /dev/null | tail -1 | cut -c1-120
done
done
for v in lib_alt lib; do
YV4_LIB_PATH=$PWD/mmdet-yolov4_amd/$v/libyv4_hip.so timeout 300 python tools/train_bench.py --batch 32 --steps 6 --warmup 3 2>&1 < /dev/null | tail -1 | cut -c1-120
done
