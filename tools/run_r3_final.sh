#!/bin/bash
# Round-3 final pass (GPU box): the -m gpu suite, the profiles of every configuration (tools/run_prof_r03.sh), the
# un-profiled bench lines (tools/run_r3_check.sh prints them).
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
bash tools/run_prof_r03.sh f32 bf16 cfg3 train > gpurun_out/r3_prof_final.log 2>&1; echo "prof rc=$?"; tail -3 gpurun_out/r3_prof_final.log
