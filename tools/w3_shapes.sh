#!/bin/bash
# the 16-bit wide 3x3 kernel's tile shapes per layer (forced: YV4_HTILE_W3x3_SHAPE(i) = 13 + 8 i), the ping-pong kernel (4) and the
# automatic choice, batch 32 and 64
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_h16.py -x -q -k "wide3x3 or w3" 2>&1 | tail -2
for B in 32 64; do
echo "=== batch $B"
python tools/conv_bench.py --dtype bf16 --batch $B --filter k3s1 --tiles 4,13,21,29,37,45,53,61 --reps 5 2>/dev/null | grep "@" | grep -v "3->32\|32->64\|64->64"
done
