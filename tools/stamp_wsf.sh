# per-phase cycle stamps of the fp32 weight-stationary 1x1 kernel (measurement build, YV4_WSF_STAMP=1):
# two workgroups print, per wave, the cycles spent in stage issue / counted wait / fragment reads + MFMAs / epilogue
source "$(dirname "${BASH_SOURCE[0]}")/_measure_lib.sh"
for f in "256->128 k1s1 @76" "128->128 k1s1 @76" "256->256 k1s1 @38"; do
  echo "== $f (tiles: ws_1x1, dma128x64, wide)"
  python tools/conv_bench.py --filter "$f" --tiles 9,6,11 --reps 5 2>&1 | grep -v amdgpu
  YV4_WSF_STAMP=1 python tools/conv_bench.py --filter "$f" --tiles 9 --reps 1 2>&1 | grep '^wsf' | sort | uniq | tail -16
done
