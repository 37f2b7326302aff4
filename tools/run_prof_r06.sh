#!/bin/bash
# Round-6 profiles (run on the GPU box through gpurun), every configuration from the FINAL tree:
#   f32   YOLOv4-L 608 b32 fp32 inference (headline)      bf16  the same in bf16
#   cfg3  YOLOv4-S 416 b256 fp16 inference (configs[3])    train YOLOv4-L 608 b64 bf16 train step (configs[2])
#   v5    YOLOv5-L 640 b64 bf16 train step (configs[4])   busy  matrix-pipe busy share of the train step's top kernels
# Raw output -> gpurun_out/prof_r06/<name>/, summaries -> gpurun_out/prof_r06/summary/ (copied into profiles/ as r06_*).
# Kernel statistics are taken over the TIMED steps only (tools/summarize_prof.py timed_region_stats: the last N repetitions
# of the launch sequence in the kernel trace), so calibration / warm-up launches no longer head the 16-bit tables.
# Counters are collected in their own runs (--pmc without any trace domain), one counter per pass as the
# microarchitecture guide prescribes; the profiled program comes directly after `--`.
# usage: tools/run_prof_r06.sh [parts]   (default: f32 bf16 cfg3 train)
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT="$GRAFT_REPO_ROOT/gpurun_out/prof_r06"
mkdir -p "$OUT"
PARTS="${*:-f32 bf16 cfg3 train}"
STEPS=10
COMMON="--steps $STEPS --warmup 2 --no-cpu-baseline --no-train --no-output-check"

prof_bench () {   # name, bench arguments...
  local name="$1"; shift
  rm -rf "$OUT/$name"; mkdir -p "$OUT/$name"
  echo $STEPS > "$OUT/$name/steps.txt"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name/trace" -- python3 bench.py $COMMON "$@" > "$OUT/$name/trace.log" 2>&1 < /dev/null
  grep '^{"metric"' "$OUT/$name/trace.log" | tail -1 > "$OUT/$name/bench_profiled.json"
  echo "$name: trace done"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/$name/pmc_fetch" -- python3 bench.py $COMMON "$@" > "$OUT/$name/pmc_fetch.log" 2>&1 < /dev/null
  grep '^{"metric"' "$OUT/$name/pmc_fetch.log" | tail -1 > "$OUT/$name/bench_pmc.json"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/$name/pmc_write" -- python3 bench.py $COMMON "$@" > "$OUT/$name/pmc_write.log" 2>&1 < /dev/null
  echo "$name: pmc done"
}

prof_train () {   # name, train_bench arguments...
  local name="$1"; shift
  rm -rf "$OUT/$name"; mkdir -p "$OUT/$name"
  echo 4 > "$OUT/$name/steps.txt"
  # one stream under the profiler: with the weight gradients on their side stream (the default) the launch ORDER in the trace
  # differs from step to step and no step period can be cut out of it; the kernels and their durations are the same
  export YV4_WGRAD_STREAM=0
  TB="tools/train_bench.py --batch 64 --steps 4 --warmup 3 --dtype bf16 $*"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name/trace" -- python3 $TB > "$OUT/$name/trace.log" 2>&1 < /dev/null
  grep '^{"metric"' "$OUT/$name/trace.log" | tail -1 > "$OUT/$name/bench_profiled.json"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/$name/pmc_fetch" -- python3 $TB > "$OUT/$name/pmc_fetch.log" 2>&1 < /dev/null
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/$name/pmc_write" -- python3 $TB > "$OUT/$name/pmc_write.log" 2>&1 < /dev/null
  unset YV4_WGRAD_STREAM
  # the step as bench.py times it -- weight gradients on their side stream: no launch period exists, the last steps are cut
  # at the optimizer's once-per-step kernel (tools/summarize_prof.py marker_region_stats)
  local side="${name}_sidestream"
  rm -rf "$OUT/$side"; mkdir -p "$OUT/$side"
  echo 4 > "$OUT/$side/steps.txt"; echo sgd_step > "$OUT/$side/marker.txt"
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/$side/trace" -- python3 $TB > "$OUT/$side/trace.log" 2>&1 < /dev/null
  grep '^{"metric"' "$OUT/$side/trace.log" | tail -1 > "$OUT/$side/bench_profiled.json"
  echo "$name: done"
}

for part in $PARTS; do
  case "$part" in
    f32)  prof_bench yolov4l_608_b32_f32 ;;
    bf16) prof_bench yolov4l_608_b32_bf16 --dtype bf16 ;;
    cfg3) prof_bench yolov4s_416_b256_f16 --model yolov4s --size 416 --batch 256 --dtype f16 ;;
    train) prof_train train_yolov4l_608_b64_bf16 ;;
    v5)   prof_train train_yolov5l_640_b64_bf16 --model yolov5l --size 640 ;;
    busy)
      # matrix-pipe busy share of the train step's kernels: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 pipes)
      name=train_busy
      rm -rf "$OUT/$name"; mkdir -p "$OUT/$name"
      rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d "$OUT/$name/sq" -- python3 tools/train_bench.py --batch 64 --steps 2 --warmup 2 --dtype bf16 > "$OUT/$name/sq.log" 2>&1 < /dev/null
      python3 - "$OUT/$name" <<'PY'
import collections, csv, glob, json, sys
sys.path.insert(0, 'tools')
d = sys.argv[1]
by = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + '/sq/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'yv4' not in r['Kernel_Name']:
            continue
        by[r['Kernel_Name'].split('(')[0][-60:]][r['Counter_Name']].append(float(r['Counter_Value']))
rows = []
for k, cs in by.items():
    g = sum(cs.get('GRBM_GUI_ACTIVE', [0.0]))
    m = sum(cs.get('SQ_VALU_MFMA_BUSY_CYCLES', [0.0]))
    w, wc = sum(cs.get('SQ_WAIT_ANY', [0.0])), sum(cs.get('SQ_WAVE_CYCLES', [0.0]))
    if g > 0:
        rows.append(dict(kernel=k, launches=len(cs['GRBM_GUI_ACTIVE']), gui_active_per_xcd=g / 8, mfma_busy=m,
                         busy_share=m / (g / 8 * 1024.0), wait_any_share=(w / wc if wc else None)))
rows.sort(key=lambda r: -r['gui_active_per_xcd'])
json.dump(rows, open(d + '/../summary_train_busy.json', 'w'), indent=1)
for r in rows[:16]:
    print(r)
PY
      ;;
  esac
done
python3 tools/summarize_prof.py --r03 "$OUT" "$OUT/summary"
[ -f "$OUT/summary_train_busy.json" ] && cp "$OUT/summary_train_busy.json" "$OUT/summary/train_busy.json"
ls -la "$OUT/summary"
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*.db" -delete
