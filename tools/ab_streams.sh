# bench.py with the step as one plan (--streams 1, the default) and as two half-batch plans on two streams, same box
for i in 1 2; do
for S in 1 2; do
echo -n "streams $S fp32: "; python bench.py --streams $S --steps 20 --warmup 5 --no-cpu-baseline --no-train 2>/dev/null | python tools/last_json.py roofline.frac roofline.all_convs_frac output_check
echo -n "streams $S bf16: "; python bench.py --streams $S --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-train 2>/dev/null | python tools/last_json.py roofline.frac output_check
echo -n "streams $S cfg3: "; python bench.py --streams $S --model yolov4s --size 416 --batch 256 --dtype f16 --steps 20 --warmup 5 --no-cpu-baseline --no-train 2>/dev/null | python tools/last_json.py roofline.frac output_check
done; done
