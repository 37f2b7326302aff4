#!/usr/bin/env python
"""Print selected fields of the last JSON line on stdin (bench.py / train_bench.py output): value, ms_per_step and the
given dotted paths.  usage: python bench.py ... | python tools/last_json.py [roofline.frac ...]"""
import json
import sys

lines = [l for l in sys.stdin if l.startswith('{')]
if not lines:
    print('no JSON line')
    sys.exit(1)
d = json.loads(lines[-1])
out = [d.get('value'), d.get('ms_per_step')]
for path in sys.argv[1:]:
    v = d
    for k in path.split('.'):
        v = v.get(k) if isinstance(v, dict) else None
    out.append(str(v)[:100])
print(*out)
