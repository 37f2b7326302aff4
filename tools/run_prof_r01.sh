#!/bin/bash
# Round-1 profiles (run on the GPU box through gpurun).  Raw output -> gpurun_out/prof_r01/,
# summaries -> gpurun_out/prof_r01/summary/ (copied into profiles/ by tools/summarize_prof.py).
export TMPDIR=/tmp
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r01
rm -rf $OUT; mkdir -p $OUT
BENCH="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train"
# 1. per-kernel time of the bench command (same command as the headline run, fewer steps)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $BENCH > $OUT/trace.log 2>&1
# 2. HBM traffic counters, one pass each (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $BENCH > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_sq -- python3 $BENCH > $OUT/pmc_sq.log 2>&1
# 3. un-profiled runs: layer table + the headline JSON line (with the CPU baseline)
python3 tools/conv_bench.py --tiles 3,5,6,7 > $OUT/conv_shapes.txt 2>&1
python3 bench.py --steps 20 --warmup 5 --no-train --layers $OUT/layers.json > $OUT/bench.json 2> $OUT/bench.err
tail -2 $OUT/bench.json
# 4. training step (row f-1): per-kernel time + the un-profiled train-step line
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_train -- python3 tools/train_bench.py --batch 32 --steps 2 --warmup 1 > $OUT/trace_train.log 2>&1
python3 tools/train_bench.py --batch 32 --steps 5 2> $OUT/train_bench.err | tail -1 > $OUT/train_bench.json
# 5. 16-bit rows (configs[2], [3]): bf16 inference bench line, bf16 train step at batch 64 + its kernel stats
python3 bench.py --dtype bf16 --no-cpu-baseline --no-train --layers $OUT/layers_bf16.json 2> $OUT/bench_bf16.err | tail -1 > $OUT/bench_bf16.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_train_bf16 -- python3 tools/train_bench.py --batch 64 --steps 2 --warmup 1 --dtype bf16 > $OUT/trace_train_bf16.log 2>&1
python3 tools/train_bench.py --batch 64 --steps 5 --dtype bf16 2> $OUT/train_bench_bf16.err | tail -1 > $OUT/train_bench_bf16.json
python3 tools/conv_bench.py --dtype bf16 --tiles 1,2,3 > $OUT/conv_shapes_bf16.txt 2>&1
python3 tools/summarize_prof.py $OUT $OUT/summary
ls -la $OUT/summary
# raw per-dispatch traces are large and already condensed: keep the pull small
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
