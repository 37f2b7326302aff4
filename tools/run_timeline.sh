#!/bin/bash
# One-step launch-order timeline of the bf16 train step (run on the GPU box through gpurun): rocprofv3 kernel trace of
# tools/train_bench.py, reduced by tools/train_timeline.py to gpurun_out/tl/<name>.txt (the raw trace is deleted).
# usage: tools/run_timeline.sh [name] [train_bench arguments...]
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
NAME="${1:-timeline}"; shift || true
ARGS="${*:---batch 64 --dtype bf16}"
OUT="$GRAFT_REPO_ROOT/gpurun_out/tl"
mkdir -p "$OUT"; rm -rf "$OUT/trace"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 tools/train_bench.py --steps 2 --warmup 2 $ARGS > "$OUT/$NAME.run.log" 2>&1 < /dev/null
python3 tools/train_timeline.py "$OUT/trace" "$OUT/$NAME.txt"
rm -rf "$OUT/trace"
