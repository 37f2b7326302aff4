# the train step with the gradient exchange ACTIVE on a one-rank RCCL group (every bucket's all-reduce is launched from the
# backward hooks, as on a node; with one rank it is the identity): weight gradients on the side stream off / on
export YV4_REDUCER_AT_WORLD1=1 YV4_EXCHANGE_AT_WORLD1=1 YV4_DIST_FORCE_INIT=1 MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for i in 1 2; do
for W in 0 1; do
for mode in allreduce direct; do
export YV4_WGRAD_STREAM=$W MASTER_PORT=$((29700 + RANDOM % 200))
echo -n "wgrad side stream $W, exchange $mode: "; python tools/train_bench.py --batch 64 --steps 8 --warmup 3 --dtype bf16 --grad-exchange $mode 2>/dev/null | python tools/last_json.py backend grad_exchange
done; done; done
unset YV4_REDUCER_AT_WORLD1 YV4_EXCHANGE_AT_WORLD1 YV4_DIST_FORCE_INIT
echo -n "no exchange, side stream 1: "; YV4_WGRAD_STREAM=1 python tools/train_bench.py --batch 64 --steps 8 --warmup 3 --dtype bf16 2>/dev/null | python tools/last_json.py
