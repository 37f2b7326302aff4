#!/bin/bash
# per-kernel time of the bf16 train step (kernel trace + stats), summary into gpurun_out/<name>_kernel_stats.csv
# usage (through gpurun): tools/prof_train_quick.sh NAME [train_bench arguments...]
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
NAME="$1"; shift
OUT="$GRAFT_REPO_ROOT/gpurun_out/prof_$NAME"
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 tools/train_bench.py --batch 64 --steps 5 --warmup 2 --dtype bf16 "$@" > "$OUT/trace.log" 2>&1 < /dev/null
grep '^{"metric"' "$OUT/trace.log" | tail -1 > "$GRAFT_REPO_ROOT/gpurun_out/${NAME}_bench_profiled.json" || true
f=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
cp "$f" "$GRAFT_REPO_ROOT/gpurun_out/${NAME}_kernel_stats.csv"
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*.db" -delete
head -25 "$GRAFT_REPO_ROOT/gpurun_out/${NAME}_kernel_stats.csv" | cut -c1-150
