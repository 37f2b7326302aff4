#!/bin/bash
# Steady-state kernel table: the bench command WITHOUT the per-box tile autotune (the fixed heuristic picks the same
# tile classes), so every launch rocprofv3 averages is a launch of the timed step and the per-kernel average can be
# compared 1:1 with the `roofline.avg_launch_us` that the same (profiled) process prints.
# Raw output -> gpurun_out/prof_steady/, summaries -> gpurun_out/prof_steady/summary/ (copy into profiles/).
export TMPDIR=/tmp
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_steady
rm -rf $OUT; mkdir -p $OUT/summary
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-train --no-autotune > $OUT/trace.log 2>&1 < /dev/null
python3 tools/summarize_prof.py $OUT $OUT/summary > /dev/null
mv $OUT/summary/kernel_stats.csv $OUT/summary/steady_kernel_stats.csv
grep '^{"metric"' $OUT/trace.log | tail -1 > $OUT/summary/steady_bench_profiled.json
rm -f $OUT/summary/pmc_per_kernel.json
python3 - $OUT/summary <<'PY'
import csv, json, sys, os
d = sys.argv[1]
b = json.load(open(os.path.join(d, 'steady_bench_profiled.json')))['roofline']
tile = b['kernel'].split('<dma')[1].rstrip('>').split('x')
rows = {r['Name']: r for r in csv.DictReader(open(os.path.join(d, 'steady_kernel_stats.csv')))}
k = [n for n in rows if 'conv_mfma_f32_dma_kernel<%s, %s,' % (tile[0], tile[1]) in n][0]
print('bench avg_launch_us %.2f | rocprofv3 %s AverageNs/1000 %.2f (%s calls)'
      % (b['avg_launch_us'], k, float(rows[k]['AverageNs']) / 1000, rows[k]['Calls']))
PY
find $OUT -name "*kernel_trace.csv" -delete
head -6 $OUT/summary/steady_kernel_stats.csv
