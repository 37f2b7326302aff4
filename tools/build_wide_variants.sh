# builds lib_var/libyv4_<name>.so for each "name:flags" argument: the two 16-bit wide kernels compiled with the flags, linked
# with the product's other objects.   bash tools/build_wide_variants.sh rd1:-DYV4_WIDE_RD1 pickold:-DYV4_WIDE_PICK_OLD
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cd $ROOT/mmdet-yolov4_amd/csrc
mkdir -p ../lib_var
OBJS=$(ls ../lib/*.o | grep -v "conv3x3_wide_h16.o\|conv_wide_h16.o")
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}; flags=${flags//,/ }
  ( for f in conv3x3_wide_h16 conv_wide_h16; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. $flags -c $f.hip -o /tmp/${f}_$name.o 2>/dev/null || exit 1; done
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS /tmp/conv3x3_wide_h16_$name.o /tmp/conv_wide_h16_$name.o -o ../lib_var/libyv4_$name.so && echo built $name ) &
done
wait
