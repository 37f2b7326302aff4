# same-box A/B of the round-4 wide-tile kernels (measurement build: YV4_W3 / YV4_WIDE switch them off)
source "$(dirname "$0")/_measure_lib.sh"
for i in 1 2; do
for w in "0 0" "1 0" "1 1"; do
set -- $w
echo -n "bf16 inf W3=$1 WIDE=$2: "; YV4_W3=$1 YV4_WIDE=$2 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-output-check 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], {k:(v['launches_per_step'],v['avg_launch_us']) for k,v in d['roofline']['tiles'].items()})"
done; done
for w in "0 0" "1 0" "1 1"; do set -- $w; echo -n "cfg3 W3=$1 WIDE=$2: "; YV4_W3=$1 YV4_WIDE=$2 python bench.py --model yolov4s --size 416 --batch 256 --dtype f16 --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-output-check 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])"; done
for w in "0 0" "1 0" "1 1"; do set -- $w; echo -n "train W3=$1 WIDE=$2: "; YV4_W3=$1 YV4_WIDE=$2 python tools/train_bench.py --batch 64 --dtype bf16 --steps 10 --warmup 3 2>&1 | tail -1 | cut -c50-110; done
