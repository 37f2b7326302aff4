# compile-time ablation of the wide 3x3 kernel's K loop (lib_var/libyv4_w3_abl<bits>.so from tools/build_w3_variants.sh; wrong
# results on purpose).  Bits: 1 no weight DMA in the loop, 128 no image DMA, 2 no MFMA, 4 no barrier, 8 no fragment reads,
# 16 no epilogue
for ab in ${ABLS:-0 2 8 10 129 131 139 143 155 16}; do
  export YV4_LIB_PATH=$PWD/mmdet-yolov4_amd/lib_var/libyv4_w3_abl$ab.so
  echo "== YV4_W3_ABL=$ab"
  python tools/conv_bench.py --dtype bf16 --batch 32 --filter k3s1 --tiles 5 --reps 7 --chain 3 2>&1 | grep -E "^(128->128|256->256|512->1024)" | cut -c1-80
done
