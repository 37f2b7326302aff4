timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fullsize_cfgs.py tests/test_gpu_v3.py -x -q -k "nms or decode or post or fullsize or cfg3 or head or v3" > gpurun_out/r4_post_tests6.log 2>&1; tail -2 gpurun_out/r4_post_tests6.log
for a in "--dtype bf16" "--dtype bf16 --graph" "--model yolov4s --size 416 --batch 256 --dtype f16" "--model yolov4s --size 416 --batch 256 --dtype f16 --graph" ""; do
echo -n "bench $a: "; python bench.py $a --steps 20 --warmup 5 --no-cpu-baseline --no-train 2>/dev/null | python tools/last_json.py output_check
done
