#!/bin/bash
# Per-layer HBM traffic table (run on the GPU box through gpurun) -> gpurun_out/pmc_layers_<dtype>/layer_traffic.md
# YV4_PROF_DTYPE=bf16 profiles the 16-bit path.
export TMPDIR=/tmp
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
DT=${YV4_PROF_DTYPE:-f32}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_layers_$DT
if [ "$DT" != f32 ]; then export YV4_ESIZE=2; fi
rm -rf $OUT; mkdir -p $OUT
BENCH="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train --dtype $DT"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $BENCH > $OUT/pmc_fetch.log 2>&1 < /dev/null
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $BENCH > $OUT/pmc_write.log 2>&1 < /dev/null
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train --dtype $DT --layers $OUT/layers.json > $OUT/bench.json 2> $OUT/bench.err < /dev/null
python3 tools/pmc_per_layer.py $OUT/pmc_fetch $OUT/pmc_write $OUT/layers.json $OUT/layer_traffic.md
find $OUT -name "*counter_collection.csv" -delete
