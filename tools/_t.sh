timeout 1500 python -m pytest tests -x -q -m gpu -k "train or syncbn or ddp or h16 or v3 or loss or hooks" 2>&1 < /dev/null | tail -2
for i in 1 2 3; do
timeout 300 python tools/train_bench.py --dtype bf16 --batch 64 --steps 8 --warmup 3 2>&1 < /dev/null | tail -1 | cut -c1-120
done
timeout 300 python tools/train_bench.py --batch 32 --steps 6 --warmup 3 2>&1 < /dev/null | tail -1 | cut -c1-120
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -- python3 tools/train_bench.py --dtype bf16 --batch 64 --steps 3 --warmup 2 > /tmp/lb.txt 2>&1 < /dev/null
f=$(find /tmp/pb -name "*kernel_stats.csv" | head -1); if [ -n "$f" ]; then grep -i "copyBuffer\|index_elementwise" "$f" | cut -c1-50,150-230; fi
