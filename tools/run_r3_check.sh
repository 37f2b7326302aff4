#!/bin/bash
# Round-3 regression + numbers (GPU box): the -m gpu suite, then the un-profiled bench lines of every configuration.
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3_check; mkdir -p $OUT
timeout -k 10 700 python -m pytest tests -m gpu -q > $OUT/suite.log 2>&1; echo "suite rc=$?"; tail -4 $OUT/suite.log
timeout -k 10 120 python bench.py --dtype bf16 --no-train --no-cpu-baseline --layers $OUT/layers_bf16.json 2> $OUT/bench_bf16.err | tail -1 > $OUT/bench_bf16.json
timeout -k 10 120 python bench.py --model yolov4s --size 416 --batch 256 --dtype f16 --no-cpu-baseline --no-train --layers $OUT/layers_cfg3.json 2> /dev/null | tail -1 > $OUT/cfg3_yolov4s_f16_416_b256.json
timeout -k 10 200 python tools/train_bench.py --batch 64 --steps 8 --warmup 3 --dtype bf16 --overlap-report 2> $OUT/train.err | tail -1 > $OUT/train_bench_bf16.json
timeout -k 10 200 python tools/train_bench.py --batch 64 --size 640 --model yolov5l --steps 8 --warmup 3 --dtype bf16 2> /dev/null | tail -1 > $OUT/cfg4_yolov5l_train_bf16_640_b64.json
python - <<PY
import json
for n in ('bench_bf16','cfg3_yolov4s_f16_416_b256','train_bench_bf16','cfg4_yolov5l_train_bf16_640_b64'):
    try:
        d=json.load(open('$OUT/'+n+'.json')); print(n, d['value'], d['ms_per_step'], d.get('output_check',''), d.get('overlap',''))
    except Exception as e: print(n,'ERR',e)
PY
