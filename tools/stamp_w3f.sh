# s_memtime stamps and in-kernel clock of the fp32 wide 3x3 kernel
# (lib_var/libyv4_w3f_stamp.so: bash tools/build_src_variants.sh w3f_stamp:conv3x3_wide_f32:-DYV4_W3F_STAMP)
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
export YV4_LIB_PATH=$PWD/mmdet-yolov4_amd/lib_var/libyv4_w3f_stamp.so
python tools/stamp_w3f.py --cin 256 --cout 256 --hw 38 2>&1 | grep -v amdgpu.ids
python tools/stamp_w3f.py --cin 128 --cout 128 --hw 76 2>&1 | grep -v amdgpu.ids
python tools/stamp_w3f.py --cin 512 --cout 1024 --hw 19 2>&1 | grep -v amdgpu.ids
