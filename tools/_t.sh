timeout 1200 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train_parity.py tests/test_gpu_train_hooks.py tests/test_gpu_ddp.py tests/test_gpu_syncbn.py tests/test_gpu_v3.py -x -q 2>&1 < /dev/null | tail -2
for i in 1 2; do
timeout 300 python tools/train_bench.py --dtype bf16 --batch 64 --steps 8 --warmup 3 2>&1 < /dev/null | tail -1 | cut -c1-120
done
