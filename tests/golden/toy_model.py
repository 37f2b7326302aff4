"""Toy model + data of the hook fixtures (tests/golden/hooks.npz): shared by the generator
(make_golden_hooks.py, which drives the REFERENCE's hooks with it) and by the tests."""
import torch
import torch.nn as nn


class Toy(nn.Module):
    """conv -> BN -> conv(bias): parameter names end in .weight/.bias, BN contributes float and
    int64 buffers -- the state-entry kinds the EMA hook distinguishes."""

    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(4, 8, 3, padding=1, bias=False)
        self.bn = nn.BatchNorm2d(8, eps=1e-3, momentum=0.03)
        self.pred = nn.Conv2d(8, 4, 1, bias=True)

    def forward(self, x):
        return self.pred(torch.nn.functional.leaky_relu(self.bn(self.conv(x)), 0.1))

    def train_step(self, data, optimizer):
        out = self(data['img'])
        loss = ((out - data['target']) ** 2).mean() * 30.0
        return dict(loss=loss, log_vars=dict(loss=float(loss.detach())), num_samples=data['img'].shape[0])


def toy_groups(model, lr, wd):
    """mmcv DefaultOptimizerConstructor with paramwise_cfg(bias_decay_mult=0, norm_decay_mult=0):
    one group per parameter in named_parameters order."""
    groups = []
    for mod_ in model.modules():
        is_norm = isinstance(mod_, nn.modules.batchnorm._BatchNorm)
        for name, p in mod_.named_parameters(recurse=False):
            g = {'params': [p]}
            if name == 'bias' and not is_norm:
                g['lr'] = lr * 1.0
            if is_norm or name == 'bias':
                g['weight_decay'] = 0.0
            groups.append(g)
    return groups


def toy_batches(n_iters, batch, seed):
    gen = torch.Generator().manual_seed(seed)
    return [dict(img=torch.randn(batch, 4, 6, 6, generator=gen), target=torch.randn(batch, 4, 6, 6, generator=gen))
            for _ in range(n_iters)]


