#!/bin/bash
# `train_bench.py --overlap-report` with TWO ranks on a one-GPU box (both on cuda:0, gloo group): where in backward every
# gradient bucket becomes exchangeable, how much of the exchange backward can hide, the exposed megabytes.  The exchange itself
# runs over gloo here (host copies), so the step time is NOT an RCCL number; the launch points are what a node would see.
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/overlap2; mkdir -p $OUT
export WORLD_SIZE=2 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29633 YV4_DIST_BACKEND=gloo
ARGS="--batch 32 --steps 4 --warmup 2 --dtype bf16 --overlap-report $@"
RANK=1 timeout -k 10 500 python3 tools/train_bench.py $ARGS > $OUT/rank1.log 2>&1 &
P1=$!
RANK=0 timeout -k 10 500 python3 tools/train_bench.py $ARGS > $OUT/rank0.log 2>&1
RC0=$?
wait $P1; RC1=$?
echo "rank0 exit $RC0, rank1 exit $RC1"
grep '^{' $OUT/rank0.log | tail -1
[ $RC0 -eq 0 ] && [ $RC1 -eq 0 ]
