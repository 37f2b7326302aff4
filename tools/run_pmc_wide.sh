#!/bin/bash
# matrix-pipe busy share of the 16-bit wide 3x3 kernel against the ping-pong kernel, layer by layer (GPU box)
export TMPDIR=/tmp
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_wide
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/sq -- python3 tools/conv_bench.py --dtype bf16 --batch 32 --filter "k3s1" --tiles 4,5 --reps 1 --warm-ms 0 > $OUT/sq.log 2>&1 < /dev/null
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sqf -- python3 tools/conv_bench.py --dtype f32 --batch 32 --filter "k3s1" --tiles 7,10 --reps 1 --warm-ms 0 > $OUT/sqf.log 2>&1 < /dev/null
python3 - <<PY
import csv,glob,collections
for d in ('sq','sqf'):
    fs=glob.glob('$OUT/'+d+'/*/*_counter_collection.csv')
    if not fs: print(d,'no csv'); continue
    by=collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        if 'conv' not in r['Kernel_Name']: continue
        k=(int(r['Dispatch_Id']), r['Kernel_Name'].split('(')[0][-44:], r['Grid_Size'])
        by.setdefault(k,{})[r['Counter_Name']]=float(r['Counter_Value'])
    seen=set()
    for k,v in by.items():
        g=v.get('GRBM_GUI_ACTIVE',0)
        if g<=0: continue
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; busy share = MFMA busy cycles / (wall cycles x 1 024 SIMDs)
        key=(k[1],k[2],round(v.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/1e6,1))
        if key in seen: continue
        seen.add(key)
        print(d, k[1], k[2], {a:round(b) for a,b in v.items()}, 'busy share=%.3f' % (v.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(g/8.0)/1024.0))
PY
