import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.argv = ['train_bench.py', '--dtype', 'bf16', '--batch', '8', '--steps', '2', '--warmup', '2']
import importlib.util
spec = importlib.util.spec_from_file_location('tb', os.path.join(os.path.dirname(__file__), 'train_bench.py'))
tb = importlib.util.module_from_spec(spec)
spec.loader.exec_module(tb)
from torch.profiler import profile, ProfilerActivity
import traceback
counts = collections.Counter()
orig_copy = torch.Tensor.copy_
# count python-level call sites of ops that end in a device copy: patch a few entry points
import torch.utils._python_dispatch as pd
class Mode(pd.TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in ('copy_', '_to_copy', 'clone', 'fill_', 'zero_', 'add_.', 'add.', 'cat')):
            st = traceback.extract_stack(limit=12)
            site = next((f'{os.path.basename(f.filename)}:{f.lineno}' for f in reversed(st)
                         if 'mmdet-yolov4_amd' in f.filename or 'train_bench' in f.filename), '?')
            counts[(name.replace('aten.', ''), site)] += 1
        return func(*args, **(kwargs or {}))
state = {'n': 0}
_main = tb.main
with Mode():
    _main()
for (name, site), c in counts.most_common(40):
    print('%6d  %-22s %s' % (c, name, site))
