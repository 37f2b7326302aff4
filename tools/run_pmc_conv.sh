#!/bin/bash
# PMC passes over one conv shape (GPU box).  Usage: run_pmc_conv.sh "<filter>" "<tiles>" <tag>
export TMPDIR=/tmp
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$3
mkdir -p $OUT
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/l2 -- python3 tools/conv_bench.py --filter "$1" --tiles $2 --reps 1 > $OUT/l2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 tools/conv_bench.py --filter "$1" --tiles $2 --reps 1 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_EA0_RDREQ_sum --output-format csv -d $OUT/write -- python3 tools/conv_bench.py --filter "$1" --tiles $2 --reps 1 > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/sq -- python3 tools/conv_bench.py --filter "$1" --tiles $2 --reps 1 > $OUT/sq.log 2>&1
python3 - <<PY
import csv,glob,collections
for d in ('l2','fetch','write','sq'):
    fs=glob.glob('$OUT/'+d+'/*/*_counter_collection.csv')
    if not fs: print(d,'no csv'); continue
    by=collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        if 'conv_' not in r['Kernel_Name']: continue
        k=(int(r['Dispatch_Id']), r['Kernel_Name'].split('<')[0][-24:]+'<'+r['Kernel_Name'].split('<')[1][:14], r['Grid_Size'])
        by.setdefault(k,{})[r['Counter_Name']]=float(r['Counter_Value'])
    for k,v in by.items(): print(d,k,{a:round(b) for a,b in v.items()})
PY
