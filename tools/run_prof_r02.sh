#!/bin/bash
# Round-2 profiles (run on the GPU box through gpurun).  Raw output -> gpurun_out/prof_r02/, summaries ->
# gpurun_out/prof_r02/summary/ (copied into profiles/ as r02_*).  Counters are collected in their own runs
# (--pmc without any trace domain), one pass per counter group as the microarchitecture guide prescribes.
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT="$GRAFT_REPO_ROOT/gpurun_out/prof_r02"
rm -rf "$OUT"; mkdir -p "$OUT"
BENCH="bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-train"
# 1. per-kernel time of the headline command (steady state: every profiled launch is a launch of the timed step)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $BENCH > $OUT/trace.log 2>&1 < /dev/null
grep '^{"metric"' $OUT/trace.log | tail -1 > $OUT/bench_profiled.json
echo "trace done"
# 2. HBM traffic of the same command, one counter per pass
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $BENCH > $OUT/pmc_fetch.log 2>&1 < /dev/null
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $BENCH > $OUT/pmc_write.log 2>&1 < /dev/null
echo "pmc done"
# 3. un-profiled: headline line with the layer table and the CPU baseline; bf16 inference; batch-1 protocol lines
python3 bench.py --steps 20 --warmup 5 --layers $OUT/layers.json > $OUT/bench.json 2> $OUT/bench.err
python3 bench.py --dtype bf16 --no-cpu-baseline --no-train --layers $OUT/layers_bf16.json 2> $OUT/bench_bf16.err | tail -1 > $OUT/bench_bf16.json
python3 bench.py --batch 1 --graph --steps 200 --warmup 20 --no-cpu-baseline --no-train 2> /dev/null | tail -1 > $OUT/bench_batch1_f32.json
python3 bench.py --batch 1 --graph --dtype bf16 --steps 200 --warmup 20 --no-cpu-baseline --no-train 2> /dev/null | tail -1 > $OUT/bench_batch1_bf16.json
echo "bench lines done"
# 4. training step (configs[2]): kernel stats + the un-profiled line; configs[3] and [4] lines
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_train_bf16 -- python3 tools/train_bench.py --batch 64 --steps 2 --warmup 1 --dtype bf16 > $OUT/trace_train_bf16.log 2>&1 < /dev/null
python3 tools/train_bench.py --batch 64 --steps 8 --warmup 3 --dtype bf16 2> $OUT/train_bench_bf16.err | tail -1 > $OUT/train_bench_bf16.json
python3 tools/train_bench.py --batch 64 --size 640 --model yolov5l --steps 8 --warmup 3 --dtype bf16 2> /dev/null | tail -1 > $OUT/cfg4_yolov5l_train_bf16_640_b64.json
python3 bench.py --model yolov4s --size 416 --batch 256 --dtype f16 --no-cpu-baseline --no-train 2> /dev/null | tail -1 > $OUT/cfg3_yolov4s_f16_416_b256.json
python3 tools/aug_bench.py 2> /dev/null | tail -1 > $OUT/aug_bench.json
python3 tools/summarize_prof.py $OUT $OUT/summary
ls -la $OUT/summary
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
