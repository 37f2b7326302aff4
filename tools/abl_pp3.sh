set -u
cd "$GRAFT_REPO_ROOT"
M=mmdet-yolov4_amd/lib_alt/libyv4_hip_measure.so
for f in "256->256 k3s1" "128->128 k3s1 @76" "512->512 k3s1"; do
  echo "== $f product"; timeout -k 5 100 python tools/conv_bench.py --dtype bf16 --tiles 4,2 --filter "$f" --chain 8 2>&1 | grep -v "^$" | tail -4
  for ab in 0 1 2 4 8 16 3 6 7; do
    echo "== $f ablate $ab"; YV4_LIB_PATH=$M YV4_H16_ABLATE=$ab timeout -k 5 100 python tools/conv_bench.py --dtype bf16 --tiles 4 --filter "$f" --chain 8 2>&1 | grep -i "us\|error" | tail -2
  done
done
