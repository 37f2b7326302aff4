# same-box comparison of builds of the wide 3x3 16-bit kernel per layer and tile shape: lib_prev, lib and lib_var/libyv4_w3_<name>.so
# bash tools/ab_w3var.sh "prev cur sk1 sk2" [batch] [extra conv_bench arguments]
B=${2:-32}
for L in $1; do
case $L in
prev) export YV4_LIB_PATH=$PWD/mmdet-yolov4_amd/lib_prev/libyv4_hip_prev.so YV4_LIB_ABI_ANY=1;;
cur) unset YV4_LIB_PATH;;
*) export YV4_LIB_PATH=$PWD/mmdet-yolov4_amd/lib_var/libyv4_w3_$L.so;;
esac
echo "== $L batch $B $3  (columns: auto shape, then shapes 256x256, 192x256, 128x256, 384x128, 256x128)"
python tools/conv_bench.py --dtype bf16 --batch $B --tiles 5,13,21,29,37,45 --filter k3s1 --reps 7 $3 2>&1 | grep -v "^3->\|^32->\|^64->\|amdgpu.ids\|weighted" | sed 's/h16_w3x3: *//g;s/best=.*//' | cut -c1-200
done
