# builds lib_var/libyv4_w3_<name>.so for each "name:flags" argument: conv3x3_wide_h16.hip compiled with the flags, linked with
# the product's other objects.   bash tools/build_w3_variants.sh abl2:-DYV4_W3_ABL=2 ...
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cd $ROOT/mmdet-yolov4_amd/csrc
mkdir -p ../lib_var
OBJS=$(ls ../lib/*.o | grep -v conv3x3_wide_h16.o)
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}; flags=${flags//,/ }
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. $flags -c conv3x3_wide_h16.hip -o /tmp/w3_$name.o 2>/dev/null &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS /tmp/w3_$name.o -o ../lib_var/libyv4_w3_$name.so && echo built $name ) &
done
wait
