# ablation of stem_down (measurement build; wrong results on purpose): bit 1 no phase B, 2 no phase C, 4 no activation,
# 8 no stores, 16 no patch loads.  Reports the stem_down launch time inside configs[3] and inside YOLOv4-L bf16.
export YV4_LIB_PATH=$PWD/mmdet-yolov4_amd/lib_alt/libyv4_hip_measure.so
for ab in 0 4 1 2 3 8 16; do
  echo -n "YV4_SD_ABLATE=$ab v4s: "
  YV4_SD_ABLATE=$ab python bench.py --model yolov4s --size 416 --batch 256 --dtype f16 --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-output-check 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['roofline']['tiles']['stem_down']['avg_launch_us'])"
  echo -n "YV4_SD_ABLATE=$ab v4l: "
  YV4_SD_ABLATE=$ab python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-output-check 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['roofline']['tiles']['stem_down']['avg_launch_us'])"
done
