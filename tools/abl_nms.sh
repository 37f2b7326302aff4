#!/bin/bash
# ablation of nms_images (measurement build; wrong results on purpose): kernel time inside the bf16 step from a kernel trace
set -u
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
source "$(dirname "$0")/_measure_lib.sh"
for ab in 0 1 8; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/abl_nms/$ab; rm -rf "$OUT"; mkdir -p "$OUT"
  YV4_NMS_ABLATE=$ab rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 bench.py --dtype bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-train --no-output-check > "$OUT/log.txt" 2>&1 < /dev/null
  f=$(find "$OUT" -name '*kernel_trace.csv' | head -1)
  if [ -n "$f" ]; then python3 - "$f" $ab <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'nms_images' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
print('YV4_NMS_ABLATE='+sys.argv[2], [round(x) for x in d[-6:]])
PY
  fi
done
