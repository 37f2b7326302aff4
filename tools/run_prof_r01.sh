#!/bin/bash
# Profiles of round 1 (run on the GPU box through gpurun).  Outputs under gpurun_out/prof_r01/.
set -x
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r01
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
# 1. per-kernel time of the bench command
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_trace.log 2>&1
# 2. counters (own run, no tracing) on two representative conv shapes
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 tools/conv_bench.py --filter "k3s1 @38" --tiles 1,3,4 --reps 2 > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $OUT/pmc_inst -- python3 tools/conv_bench.py --filter "k3s1 @38" --tiles 1,3,4 --reps 2 > $OUT/pmc_inst.log 2>&1
find $OUT -name "*.csv" | head -30
