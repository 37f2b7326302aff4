# non-temporal output stores in 16-bit inference plans (yv4_conv_desc.flags, ABI 7): off / on, same box, alternating
for i in 1 2 3; do
for NT in 0 1; do
export YV4_PLAN_NT=$NT
echo -n "nt=$NT v4l bf16 inference: "; python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-output-check 2>/dev/null | python tools/last_json.py
echo -n "nt=$NT cfg3: "; python bench.py --model yolov4s --size 416 --batch 256 --dtype f16 --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-output-check 2>/dev/null | python tools/last_json.py
done; done
