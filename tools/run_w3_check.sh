# wide 3x3 kernels after a change: tests, LDS conflict counters, per-layer times, whole-network lines
timeout -k 10 900 python -m pytest tests/test_gpu_h16.py tests/test_gpu_parity.py -x -q -k "wide or w3 or W3 or bitwise" > gpurun_out/r4_w3_tests.log 2>&1; tail -2 gpurun_out/r4_w3_tests.log
bash tools/run_pmc_lds.sh "256->256 k3s1" "4,5" w3lds 2>&1 < /dev/null | grep "^lds" | cut -c1-330
python tools/conv_bench.py --dtype bf16 --batch 32 --filter k3s1 --tiles 4,5 --reps 7 --chain 3 2>&1 | grep -E "^(128->128|256->256|512->|128->256|256->512)" | cut -c1-110
python tools/conv_bench.py --dtype f32 --batch 32 --filter k3s1 --tiles 7,10 --reps 5 --chain 3 2>&1 | grep -E "^(128->128|256->256|512->|128->256|256->512)" | cut -c1-110
for a in "--dtype bf16" ""; do
echo -n "bench $a: "; python bench.py $a --steps 20 --warmup 5 --no-cpu-baseline --no-train 2>/dev/null | python tools/last_json.py output_check
done
