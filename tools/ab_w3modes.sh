# the wide 3x3 16-bit layers under the conditions a network imposes: residual in the epilogue, L2 flushed (64 MB written
# between launches) or every cache flushed (300 MB), back-to-back launches.   bash tools/ab_w3modes.sh "prev cur r0 r1"
for L in ${1:-prev cur}; do
case $L in
prev) export YV4_LIB_PATH=$PWD/mmdet-yolov4_amd/lib_prev/libyv4_hip_prev.so YV4_LIB_ABI_ANY=1;;
cur) unset YV4_LIB_PATH;;
*) export YV4_LIB_PATH=$PWD/mmdet-yolov4_amd/lib_var/libyv4_w3_$L.so;;
esac
for mode in "" "--res" "--res --flush-mb 300"; do
echo "== $L $mode"
python tools/conv_bench.py --dtype bf16 --batch 32 --tiles 5 --filter k3s1 --reps 7 $mode 2>&1 | grep -E "^(128->128|256->256|128->256|256->512|512->1024)" | sed 's/h16_w3x3: *//g;s/best=.*//' | cut -c1-60
done; done
