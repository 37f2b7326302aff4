# the fp32 weight-stationary 1x1 kernel with parts removed (measurement build; wrong results on purpose).  YV4_WSF_ABL bits:
# 1 a quarter of the output store instructions, 2 no stage DMAs, 4 no fragment reads / MFMAs, 8 no epilogue, 16 stage DMAs
# issued but out of range (zero fill: nothing fetched from HBM); --act 0 = no Mish
source "$(dirname "${BASH_SOURCE[0]}")/_measure_lib.sh"
for f in "256->128 k1s1 @76" "128->128 k1s1 @76"; do
  for v in 0 1 8 2 16 10 4 12 6 14; do
    echo -n "YV4_WSF_ABL=$v: "
    YV4_WSF_ABL=$v python tools/conv_bench.py --filter "$f" --tiles 9 --reps 7 2>&1 | grep -v 'amdgpu\|weighted'
  done
done
