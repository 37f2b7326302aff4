#!/bin/bash
# Rehearsal of `bench.py --gpus 2` on a ONE-GPU box: both ranks use cuda:0 (LOCAL_RANK=0) and line up over gloo
# (YV4_DIST_BACKEND; RCCL refuses two ranks on one device).  Exercises the launcher contract, the barrier/max
# timing, the train-step child processes and their own rendezvous port -- not the xGMI all-reduce rate.
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/rehearse2; mkdir -p $OUT
export WORLD_SIZE=2 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 YV4_DIST_BACKEND=gloo
ARGS="--gpus 2 --steps 5 --warmup 2 --batch 16 --train-batch 16 $@"
RANK=1 timeout -k 10 500 python3 bench.py $ARGS > $OUT/rank1.log 2>&1 &
P1=$!
RANK=0 timeout -k 10 500 python3 bench.py $ARGS > $OUT/rank0.log 2>&1
RC0=$?
wait $P1; RC1=$?
echo "rank0 exit $RC0, rank1 exit $RC1"
tail -2 $OUT/rank0.log
[ $RC0 -eq 0 ] && [ $RC1 -eq 0 ]
