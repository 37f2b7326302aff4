#!/bin/bash
# LDS / issue counters of the weight-gradient kernels on one layer shape (GPU box).  Usage: run_pmc_wgrad.sh "<filter>" <tag>
set -u
export TMPDIR=/tmp
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
OUT="$GRAFT_REPO_ROOT/gpurun_out/pmc_$2"
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/lds" -- python3 tools/wgrad_bench.py --det --chain 1 --filter "$1" > "$OUT/lds.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq" -- python3 tools/wgrad_bench.py --det --chain 1 --filter "$1" > "$OUT/sq.log" 2>&1
python3 - <<PY
import csv,glob,collections
for d in ('lds','sq'):
    fs=glob.glob('$OUT/'+d+'/*/*_counter_collection.csv')
    if not fs: print(d,'no csv'); continue
    by=collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        if 'wgrad' not in r['Kernel_Name'] or 'reduce' in r['Kernel_Name']: continue
        k=(r['Kernel_Name'][:70], r['Grid_Size'])
        e=by.setdefault(k,collections.defaultdict(list)); e[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in by.items(): print(d,k,{a:round(sum(b)/len(b)) for a,b in v.items()})
PY
find "$OUT" -name "*.csv" -size +2M -delete
