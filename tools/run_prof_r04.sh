#!/bin/bash
# Round-4 profiles (run on the GPU box through gpurun): per-kernel time and HBM counters of the 16-bit configurations
# (bf16 YOLOv4-L b32 inference, configs[3] fp16 YOLOv4-S b256, the bf16 train step) and of the fp32 headline.
# Raw output -> gpurun_out/prof_r04/, summaries -> gpurun_out/prof_r04/summary/ (copied into profiles/ as r04_*).
# Counters are collected in their own runs (--pmc without any trace domain), one counter per pass as the
# microarchitecture guide prescribes; the profiled program comes directly after `--`.
# usage: tools/run_prof_r04.sh [parts]   parts = any of: f32 bf16 cfg3 train   (default: all)
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT="$GRAFT_REPO_ROOT/gpurun_out/prof_r04"
mkdir -p "$OUT"
PARTS="${*:-f32 bf16 cfg3 train}"
COMMON="--steps 10 --warmup 2 --no-cpu-baseline --no-train --no-output-check"

prof_bench () {   # name, bench arguments...
  local name="$1"; shift
  rm -rf "$OUT/$name"; mkdir -p "$OUT/$name"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name/trace" -- python3 bench.py $COMMON "$@" > "$OUT/$name/trace.log" 2>&1 < /dev/null
  grep '^{"metric"' "$OUT/$name/trace.log" | tail -1 > "$OUT/$name/bench_profiled.json"
  echo "$name: trace done"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/$name/pmc_fetch" -- python3 bench.py $COMMON "$@" > "$OUT/$name/pmc_fetch.log" 2>&1 < /dev/null
  grep '^{"metric"' "$OUT/$name/pmc_fetch.log" | tail -1 > "$OUT/$name/bench_pmc.json"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/$name/pmc_write" -- python3 bench.py $COMMON "$@" > "$OUT/$name/pmc_write.log" 2>&1 < /dev/null
  echo "$name: pmc done"
}

for part in $PARTS; do
  case "$part" in
    f32)  prof_bench yolov4l_608_b32_f32 ;;
    bf16) prof_bench yolov4l_608_b32_bf16 --dtype bf16 ;;
    cfg3) prof_bench yolov4s_416_b256_f16 --model yolov4s --size 416 --batch 256 --dtype f16 ;;
    train)
      name=train_yolov4l_608_b64_bf16
      rm -rf "$OUT/$name"; mkdir -p "$OUT/$name"
      TB="tools/train_bench.py --batch 64 --steps 3 --warmup 2 --dtype bf16"
      rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name/trace" -- python3 $TB > "$OUT/$name/trace.log" 2>&1 < /dev/null
      grep '^{"metric"' "$OUT/$name/trace.log" | tail -1 > "$OUT/$name/bench_profiled.json"
      rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/$name/pmc_fetch" -- python3 $TB > "$OUT/$name/pmc_fetch.log" 2>&1 < /dev/null
      rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/$name/pmc_write" -- python3 $TB > "$OUT/$name/pmc_write.log" 2>&1 < /dev/null
      echo "$name: done" ;;
  esac
done
python3 tools/summarize_prof.py --r03 "$OUT" "$OUT/summary"
ls -la "$OUT/summary"
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*.db" -delete
