#!/bin/bash
# In-network durations of the wide 3x3 16-bit launches of one timed bf16 inference step, for lib_prev and lib (kernel trace only).
set -eu
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT="$GRAFT_REPO_ROOT/gpurun_out/prof_w3"
for L in ${1:-prev cur}; do
  rm -rf "$OUT/$L"; mkdir -p "$OUT/$L"
  if [ $L = prev ]; then export YV4_LIB_PATH=$PWD/mmdet-yolov4_amd/lib_prev/libyv4_hip_prev.so YV4_LIB_ABI_ANY=1; else unset YV4_LIB_PATH YV4_LIB_ABI_ANY; fi
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/$L/trace" -- python3 bench.py --dtype bf16 --steps 10 --warmup 2 --no-cpu-baseline --no-train --no-output-check > "$OUT/$L/trace.log" 2>&1 < /dev/null
  python3 tools/w3_innet_table.py "$OUT/$L/trace" > "$OUT/w3_innet_$L.txt"
  rm -rf "$OUT/$L/trace"
done
