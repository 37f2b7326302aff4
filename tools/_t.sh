export YV4_DIST_BACKEND=gloo MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 WORLD_SIZE=2 LOCAL_RANK=0
RANK=1 timeout 600 python bench.py --gpus 2 --steps 6 --warmup 2 --batch 16 > /tmp/r1.log 2>&1 < /dev/null &
P=$!
RANK=0 timeout 600 python bench.py --gpus 2 --steps 6 --warmup 2 --batch 16 2>&1 < /dev/null | tail -2 | cut -c1-700
wait $P; echo "rank1 rc=$?"; tail -2 /tmp/r1.log | cut -c1-200
