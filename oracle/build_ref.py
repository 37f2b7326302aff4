"""Build oracle/_ref/: the REFERENCE's own CPU Mish kernel, compiled from its sources where
they lie under /root/reference (never copied), plus oracle/ref_mish_binding.cpp.

Test infrastructure only.  A no-op where /root/reference is absent (the GPU box): the
prebuilt oracle/_ref/*.so travels with the repo snapshot instead.
"""
import glob
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = '/root/reference/mmdet/ops/mish_cuda/src/kernel/mish_cpu.cc'
OUT = os.path.join(HERE, '_ref')
NAME = 'mish_ref_ext'


def _existing():
    hits = glob.glob(os.path.join(OUT, NAME + '*.so'))
    return hits[0] if hits else None


def build(verbose=False):
    if not os.path.exists(REF_SRC):
        return _existing()
    so = _existing()
    binding = os.path.join(HERE, 'ref_mish_binding.cpp')
    if so and os.path.getmtime(so) >= max(os.path.getmtime(REF_SRC), os.path.getmtime(binding)):
        return so
    os.makedirs(OUT, exist_ok=True)
    from torch.utils.cpp_extension import load
    load(name=NAME, sources=[REF_SRC, binding], build_directory=OUT, extra_cflags=['-O2'], verbose=verbose,
         with_cuda=False)
    return _existing()


def load_ext():
    """Import the built extension (None if it was never built)."""
    so = _existing() or build()
    if not so:
        return None
    import torch  # noqa: F401  (must be imported before the extension)
    spec = importlib.util.spec_from_file_location(NAME, so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == '__main__':
    print(build(verbose='-v' in sys.argv))
