import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    return load


def state_dict_from(npz):
    """The checkpoint stored in a golden fixture (fp16-exact values) as fp32 torch tensors."""
    sd = {}
    for k in npz.files:
        if k.startswith('sd/'):
            a = npz[k]
            t = torch.from_numpy(a.astype(np.float32) if a.dtype == np.float16 else a)
            sd[k[3:]] = t
    return sd


def arch_from(npz):
    stages = [str(s) for s in npz['meta_stages']]
    reps = [None if r < 0 else int(r) for r in npz['meta_reps']]
    chans = [int(c) for c in npz['meta_channels']]
    return stages, reps, chans


@pytest.fixture(scope='session')
def gpu_device():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')
