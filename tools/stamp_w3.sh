# s_memtime stamps of the wide 3x3 16-bit kernel (lib_var/libyv4_w3_stamp.so: tools/build_w3_variants.sh stamp:-DYV4_W3_STAMP),
# hot (the layer re-run on the same tensors) and as a network meets it (residual, every cache flushed)
export YV4_LIB_PATH=$PWD/mmdet-yolov4_amd/lib_var/libyv4_w3_stamp.so
for mode in "" "--res --flush-mb 300"; do
python tools/stamp_w3.py --cin 256 --cout 256 --hw 38 --tile 21 $mode 2>&1 | grep -v amdgpu.ids
python tools/stamp_w3.py --cin 128 --cout 128 --hw 76 --tile 37 $mode 2>&1 | grep -v amdgpu.ids
python tools/stamp_w3.py --cin 512 --cout 1024 --hw 19 --tile 13 $mode 2>&1 | grep -v amdgpu.ids
done
