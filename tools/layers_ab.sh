#!/bin/bash
# per-conv in-network times (bench.py --layers) of the product and of a library variant, bf16 inference, same box.  VAR=name
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
V=$GRAFT_REPO_ROOT/mmdet-yolov4_amd/lib_var/libyv4_${VAR:?VAR=name}.so
python bench.py --dtype bf16 --steps 20 --warmup 5 --event-every 2 --no-cpu-baseline --no-train --no-output-check --layers gpurun_out/layers_product.json > /dev/null 2>&1
YV4_LIB_PATH=$V python bench.py --dtype bf16 --steps 20 --warmup 5 --event-every 2 --no-cpu-baseline --no-train --no-output-check --layers gpurun_out/layers_var.json > /dev/null 2>&1
python - <<'PY'
import json
a=json.load(open('gpurun_out/layers_product.json')); b=json.load(open('gpurun_out/layers_var.json'))
ra=a['layers'] if isinstance(a,dict) else a; rb=b['layers'] if isinstance(b,dict) else b
tot=0
for x,y in zip(ra,rb):
    d=x['us']-y['us']
    if x.get('tile')!=y.get('tile') or abs(d)>3:
        print(x.get('name'), f"{x['Cin']}->{x['Cout']} k{x['k']}s{x['stride']} @{x['H']}", 'product', x.get('tile'), round(x['us'],1), '| var', y.get('tile'), round(y['us'],1), 'diff', round(d,1))
    tot+=d
print('sum of differences (product - var), us:', round(tot,1))
PY
