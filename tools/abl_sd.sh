# ablation of stem_down (measurement build; wrong results on purpose): bit 1 no phase B, 2 no phase C, 4 no activation,
# 8 no stores, 16 no patch loads.  The kernel alone (tools/sd_bench.py) at configs[3]'s and YOLOv4-L's shapes.
source "$(dirname "${BASH_SOURCE[0]}")/_measure_lib.sh"
for ab in 0 4 1 2 3 8 16 5 6; do
  echo "== YV4_SD_ABLATE=$ab"
  YV4_SD_ABLATE=$ab python tools/sd_bench.py
done
for k in "YV4_SD_WAVES=8" "YV4_SD_TY=15"; do
  echo "== $k"
  env $k python tools/sd_bench.py
done
