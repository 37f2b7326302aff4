export YV4_LIB_PATH=$PWD/mmdet-yolov4_amd/lib_alt/libyv4_hip_measure.so
for i in 1 2; do
for w in 0 1; do
echo -n "bf16 inf W3=$w: "; YV4_W3=$w python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-output-check 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], {k:(v['launches_per_step'],v['avg_launch_us'],v['tflops']) for k,v in d['roofline']['tiles'].items() if '3x3' in k})"
done; done
for w in 0 1; do echo -n "train W3=$w: "; YV4_W3=$w python tools/train_bench.py --batch 64 --dtype bf16 --steps 10 --warmup 3 2>&1 | tail -1 | cut -c50-110; done
python - <<'PY'
import sys, torch
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import test_gpu_h16 as t
for shape in [(2,19,19,128,128),(9,38,38,256,256),(3,38,38,64,192)]:
    N,H,W,Ci,Co=shape
    outs=[t._h16_conv('cuda:0', torch.bfloat16, N,H,W,Ci,Co,3,1,1,act=1,tile=tl,raw=True) for tl in (2,4,5,13,37)]
    print(shape, [bool(torch.equal(outs[0],o)) for o in outs[1:]], [float((outs[0].float()-o.float()).abs().max()) for o in outs[1:]])
PY
