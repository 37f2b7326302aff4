# builds lib_var/libyv4_<name>.so for each "name:source:flags" argument: the named source of csrc/ compiled with the flags,
# linked with the product's other objects.   bash tools/build_src_variants.sh noroles:conv3x3_wide_f32:-DYV4_W3F_ROLES=0 ...
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cd $ROOT/mmdet-yolov4_amd/csrc
mkdir -p ../lib_var
for spec in "$@"; do
  name=${spec%%:*}; rest=${spec#*:}; src=${rest%%:*}; flags=${rest#*:}; flags=${flags//,/ }
  OBJS=$(ls ../lib/*.o | grep -v "/$src.o")
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. $flags -c $src.hip -o /tmp/var_$name.o 2>/dev/null &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS /tmp/var_$name.o -o ../lib_var/libyv4_$name.so && echo built $name ) &
done
wait
