# sourced by the A/B and ablation scripts: the measurement build (-DYV4_MEASURE: kernel knobs and ablation switches read from
# the environment) stays out of the gpurun push (.gpurunignore); build it on the box when it is not there (~1 min, 16 cores)
_ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
if [ ! -f "$_ROOT/mmdet-yolov4_amd/lib_alt/libyv4_hip_measure.so" ]; then
  echo "building the measurement library ..." >&2
  make -C "$_ROOT/mmdet-yolov4_amd/csrc" measure -j16 > /dev/null 2>&1 || { echo "make measure failed" >&2; exit 1; }
fi
export YV4_LIB_PATH="$_ROOT/mmdet-yolov4_amd/lib_alt/libyv4_hip_measure.so"
