"""A well-conditioned random initialisation of a PhysicsNet, for synthetic workloads.

PyTorch's default initialisation leaves the six VariableNets' raw outputs with a standard deviation of 7-14, i.e. about half of the
P / T / q / rho values of a random field start ON a clip bound of `inverse_norm` (interface_physics.py:256-261), and the reference's vapour formula
(`get_qs`, :181-185: q_s = 0.622 e_s / (p - 0.378 e_s)) can then hit an exactly cancelling denominator at one of millions of points within a few
optimiser steps -- the NaN is the reference's behaviour, which the kernels reproduce.  A benchmark that trains 61 random fields for hundreds of steps
from that state measures NaN operands (which run faster: DESIGN.md 6a).  This module draws every weight uniformly with ~1 / sqrt(fan_in) scaling (the
hyper-network heads, whose OUTPUT is a weight matrix, another factor 8 smaller; LayerNorm gains 1 +- 0.1; small biases), so the raw outputs stay O(1)
and every physical value starts well inside its clip bounds.  The scales are the ones the parity tests' closed-form fill uses; the values here come
from torch's generator (a seed), nothing is shared with the test infrastructure.
"""
import math

import torch


def _scale_for(name, shape):
    leaf = name.split('.')[-1]
    if name.endswith('learnable_token'):
        return 0.5
    if leaf == 'bias':
        return 0.05
    if leaf == 'weight':
        if len(shape) == 1:                       # LayerNorm gain: 1 + 0.1 u
            return 0.1
        fan_in = 1
        for s in shape[1:]:
            fan_in *= int(s)
        if 'coord_input_fc' in name or 'coord_hidden_fc' in name:
            return 1.7 / math.sqrt(fan_in) / 8.0
        return 1.7 / math.sqrt(fan_in)
    return 0.1


@torch.no_grad()
def scaled_init_(module, seed=1):
    """Re-draws every floating parameter of `module` in place (buffers -- the sinusoid table -- are left alone).  The same seed gives the same
    weights on every rank and device: the values are drawn on the host."""
    gen = torch.Generator(device='cpu')
    gen.manual_seed(int(seed))
    for name, p in module.named_parameters():
        if not torch.is_floating_point(p):
            continue
        u = torch.rand(p.shape, generator=gen, dtype=torch.float32).mul_(2.0).sub_(1.0).mul_(_scale_for(name, tuple(p.shape)))
        if name.split('.')[-1] == 'weight' and p.dim() == 1:
            u.add_(1.0)
        p.copy_(u.to(device=p.device, dtype=p.dtype))
    return module
