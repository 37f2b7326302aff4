# same-box A/B of the fp32 wide-tile 3x3 kernel (measurement build: YV4_W3F=0 switches it off), then the product line
source "$(dirname "$0")/_measure_lib.sh"
for i in 1 2 3; do
for w in 0 1; do
echo -n "fp32 inf W3F=$w: "; YV4_W3F=$w python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-output-check 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['all_convs_frac'], {k:(v['launches_per_step'],v['avg_launch_us'],v['tflops']) for k,v in d['roofline']['tiles'].items() if k in ('w3x3','dma128x128')})"
done; done
unset YV4_LIB_PATH
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('product:', d['value'], d['ms_per_step'], d['output_check'])"
