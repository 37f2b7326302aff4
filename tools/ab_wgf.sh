# same-box A/B of the general fp32 wide-tile kernel (measurement build: YV4_WGF=0 switches it off)
source "$(dirname "$0")/_measure_lib.sh"
for i in 1 2; do
for cfg in "YV4_WGF=0" "YV4_WGF=1" "YV4_WGF=1 YV4_WGF_MINOUT=16384" "YV4_WGF=1 YV4_WGF_MINOUT=8192 YV4_WGF_MAXWASTE=40"; do
echo -n "fp32 inf $cfg: "; env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-output-check 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline']['all_convs_frac'], {k:(v['launches_per_step'],v['avg_launch_us'],v['tflops']) for k,v in d['roofline']['tiles'].items() if k in ('wide','dma128x128','dma128x64','dma64x64')})"
done; done
