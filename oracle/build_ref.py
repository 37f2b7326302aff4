"""Build oracle/_ref/: the REFERENCE's own CPU Mish kernel, compiled from its sources where
they lie under /root/reference (never copied), plus oracle/ref_mish_binding.cpp.

Test infrastructure only.  A no-op where /root/reference is absent (the GPU box).  The built
oracle/_ref/*.so stays in THIS container (.gitignore and .gpurunignore both list oracle/_ref/): it is
used here to generate and re-check tests/golden/mish.npz and eval.npz, which are what travels and
what the GPU tests compare against.
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


# ---- the reference's Cython evaluation kernels (mmdet/ops/eval_utils/{iou,match}/*.pyx) ----------
EVAL_SRC = {'iou_coco': '/root/reference/mmdet/ops/eval_utils/iou/iou_coco.pyx',
            'match_coco': '/root/reference/mmdet/ops/eval_utils/match/match_coco.pyx'}


def _existing_eval(name):
    hits = glob.glob(os.path.join(OUT, name + '.*.so')) + glob.glob(os.path.join(OUT, name + '.so'))
    return hits[0] if hits else None


def build_eval(verbose=False):
    """cython (pyx -> C, written into oracle/_ref/) + gcc, straight from the sources where they lie."""
    import subprocess
    import sysconfig
    import numpy as np
    out = {}
    for name, src in EVAL_SRC.items():
        so = _existing_eval(name)
        if not os.path.exists(src):
            out[name] = so
            continue
        if so and os.path.getmtime(so) >= os.path.getmtime(src):
            out[name] = so
            continue
        os.makedirs(OUT, exist_ok=True)
        c_file = os.path.join(OUT, name + '.c')
        target = os.path.join(OUT, name + sysconfig.get_config_var('EXT_SUFFIX'))
        cmds = [[sys.executable, '-m', 'cython', '-3', src, '-o', c_file],
                ['gcc', '-O2', '-shared', '-fPIC', '-w', '-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION',
                 '-I' + sysconfig.get_paths()['include'], '-I' + np.get_include(), c_file, '-o', target]]
        for cmd in cmds:
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f'building the reference {name}.pyx failed:\n{r.stderr[-2000:]}')
            if verbose:
                print(' '.join(cmd))
        os.remove(c_file)
        out[name] = target
    return out


def load_eval():
    """(iou_coco, match_coco) of the compiled reference, or None where it was never built."""
    built = {n: _existing_eval(n) for n in EVAL_SRC}
    if not all(built.values()):
        built = build_eval()
    if not all(built.values()):
        return None
    fns = []
    for name in ('iou_coco', 'match_coco'):
        spec = importlib.util.spec_from_file_location(name, built[name])
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        fns.append(getattr(mod, name))
    return tuple(fns)


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
    print(build_eval(verbose='-v' in sys.argv))
