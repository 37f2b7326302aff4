"""The oracle's restatement of the optimizer-side training step (oracle/train_hooks_oracle.py)
against the fixture produced by the reference's own hooks (tests/golden/hooks.npz)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
from toy_model import Toy, toy_groups  # noqa: E402

from oracle import train_hooks_oracle as TO  # noqa: E402


@pytest.fixture(scope='module')
def G():
    return np.load(os.path.join(HERE, 'golden', 'hooks.npz'), allow_pickle=False)


def load_toy(G, device='cpu'):
    model = Toy()
    sd = {k[5:]: torch.from_numpy(G[k]) for k in G.files if k.startswith('init/')}
    model.load_state_dict(sd)
    return model.to(device)


def batches(G, n, device='cpu'):
    return [dict(img=torch.from_numpy(G[f'batch{i}/img']).to(device),
                 target=torch.from_numpy(G[f'batch{i}/target']).to(device)) for i in range(n)]


def test_schedule_known_answers(G):
    its, kat = G['kat_iters'], G['kat']
    for i, row in zip(its, kat):
        assert TO.warmup_value(int(i), 10000, 10., 1.0) == row[0]
        assert TO.warmup_value(int(i), 10000, 0., 1.0) == row[1]
        assert TO.warmup_value(int(i), 10000, 0.95, 1.0) == row[2]
        assert TO.ema_momentum(0.9999, int(i), 10000, 1) == row[3]
    for spg, world, want in G['kat_accum']:
        assert TO.accumulation_steps(64, int(spg), int(world)) == want


def test_simulated_training_matches_reference_hooks(G):
    cfg = json.loads(str(G['cfg_json']))
    model = load_toy(G)
    groups = [dict(lr=g.get('lr', cfg['lr']), momentum=cfg['momentum'],
                   weight_decay=g.get('weight_decay', cfg['weight_decay']), nesterov=cfg['nesterov'])
              for g in toy_groups(model, cfg['lr'], cfg['weight_decay'])]
    sim = TO.HookSimulator(model, groups, cfg)
    assert sim.accum == int(G['accumulation']) and sim.interval == int(G['ema_interval'])
    data = batches(G, cfg['iters_per_epoch'])
    losses = []
    for ep in range(cfg['epochs']):
        model.train()
        sim.swap()
        np.testing.assert_allclose(model.conv.weight.detach().numpy(), G[f'swap_in/{ep}/conv.weight'], rtol=1e-6,
                                   atol=1e-7)
        for b in data:
            sim.before_iter()
            out = model.train_step(b, None)
            losses.append(float(out['loss']))
            it = sim.iter
            sim.after_iter(out['loss'])
            sd = model.state_dict()
            for k, v in sd.items():
                np.testing.assert_allclose(v.detach().float().numpy(), G[f'iter{it}/{k}'], rtol=2e-6, atol=2e-7,
                                           err_msg=f'iter {it} {k}')
                np.testing.assert_allclose(sim.ema[k].float().numpy(), G[f"iter{it}/ema_{k.replace('.', '_')}"],
                                           rtol=2e-6, atol=2e-7, err_msg=f'iter {it} ema {k}')
        sim.swap()
        for k, v in model.state_dict().items():
            np.testing.assert_allclose(v.detach().float().numpy(), G[f'epoch_end{ep}/{k}'], rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(np.array(sim.lr_log), G['lr'], rtol=0, atol=0)
    np.testing.assert_allclose(np.array(sim.mom_log), G['momentum'], rtol=0, atol=0)
    np.testing.assert_allclose(np.array(losses), G['loss'], rtol=1e-5)
    np.testing.assert_allclose(np.array(sim.norm_log), G['grad_norm'], rtol=1e-5)
