cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_h16.py tests/test_gpu_train_ops.py tests/test_gpu_fullsize.py -m gpu -q -x -k "splitk or split or packed_weight or pp3" > gpurun_out/r3_t6.log 2>&1; tail -5 gpurun_out/r3_t6.log
for dt in bf16 f16; do
  timeout -k 10 200 python bench.py --batch 1 --graph --dtype $dt --steps 200 --warmup 20 --no-cpu-baseline --no-train 2>/dev/null | tail -1 > gpurun_out/r3_bench_batch1_${dt}_graph.json
  python -c "
import json; d=json.load(open('gpurun_out/r3_bench_batch1_${dt}_graph.json')); print('$dt', d['value'], d['ms_per_step'])"
done
bash tools/run_prof_r03.sh train f32 > gpurun_out/r3_prof_b.log 2>&1; tail -5 gpurun_out/r3_prof_b.log
