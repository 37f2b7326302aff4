#!/bin/bash
# ablation of decode_filter (measurement build; wrong results on purpose except 0 and 64): kernel time inside the bf16
# YOLOv4-L step and configs[3], from a kernel trace.  bits: 1 no sigmoid, 2 no candidate output, 4 no class loop,
# 8 no box stores, 32 no coordinate-max atomic
set -u
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
source "$(dirname "$0")/_measure_lib.sh"
for ab in ${ABLATIONS:-0 2}; do
  for cfg in "--dtype bf16" "--model yolov4s --size 416 --batch 256 --dtype f16"; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/abl_dec/$ab; rm -rf "$OUT"; mkdir -p "$OUT"
  YV4_DEC_ABLATE=$ab rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 bench.py $cfg --steps 6 --warmup 2 --no-cpu-baseline --no-train --no-output-check > "$OUT/log.txt" 2>&1 < /dev/null
  f=$(find "$OUT" -name '*kernel_trace.csv' | head -1)
  if [ -n "$f" ]; then python3 - "$f" "YV4_DEC_ABLATE=$ab $cfg" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'decode_filter' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
print(sys.argv[2], [round(x) for x in d[-6:]])
PY
  fi
  done
done
