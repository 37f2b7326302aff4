# ablation of the wide 3x3 kernel's K loop (measurement build; wrong results on purpose)
export YV4_LIB_PATH=$PWD/mmdet-yolov4_amd/lib_alt/libyv4_hip_measure.so
for ab in 0 2 10 131 139 143 159 16 8; do
  echo "== YV4_H16_ABLATE=$ab"
  YV4_H16_ABLATE=$ab python tools/conv_bench.py --dtype bf16 --batch 32 --filter k3s1 --tiles 5 --reps 7 --chain 3 2>&1 | grep -E "^(128->128|256->256|512->1024)" | cut -c1-80
done
