"""Initialisers and lenient checkpoint loading.

The initial statistics matter for synthetic-data runs and for the seeded whole-model parity tests, so they follow
the reference (model/utils.py:5-28): convolutions Kaiming-normal on fan-out for ReLU with zero bias, BatchNorm2d to
(1, 0), Linear layers Kaiming-normal (default fan-in) with zero bias.  `load_state_dict` is the reference's
"take what matches by name and shape" loader used for ImageNet checkpoints.
"""
import os

import torch.nn as nn
from torch.nn import init


def _zero_bias(layer):
    if getattr(layer, 'bias', None) is not None:
        init.zeros_(layer.bias)


def initialize_weights(m):
    """`module.apply(initialize_weights)`-style visitor."""
    if isinstance(m, nn.Conv2d):
        init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
        _zero_bias(m)
        return
    if isinstance(m, nn.BatchNorm2d):
        init.ones_(m.weight)
        init.zeros_(m.bias)
        return
    if isinstance(m, nn.Linear):
        init.kaiming_normal_(m.weight.data)
        _zero_bias(m)


def load_state_dict(model, state_dict):
    """Copy the entries of `state_dict` whose key exists in `model` with the same shape; keep the rest as initialised."""
    target = model.state_dict()
    usable = {}
    for key, value in state_dict.items():
        if key in target and tuple(value.shape) == tuple(target[key].shape):
            usable[key] = value
    target.update(usable)
    model.load_state_dict(target)
    return sorted(usable)


def wide_linear(layer, x):
    """`layer(x)` for the classifier / part FCs that sit directly on a pooled vector (tens of thousands of features,
    a few hundred outputs): the split-K f32-MFMA kernel hk_linear_fwd - 134 us against rocBLAS' 385 us at
    the BCNN shape (it picks a 16x64 macro-tile for a 45 us HBM-bound product), 30 vs 79 us at MPN's, 203 vs 266 us at
    OSME's; the backward takes hk_linear_bwd where that wins and the library GEMMs elsewhere (functional._Linear).  The
    `nn.Linear` stays the parameter holder, so `state_dict` keys, initialisers and optimiser groups are untouched.
    HAWKEYE_HIP_LINEAR=0 switches back to `layer(x)` (A/B lever)."""
    if os.environ.get('HAWKEYE_HIP_LINEAR', '1') != '0':
        from .. import functional as F        # hawkeye_amd.functional
        return F.linear(x, layer.weight, layer.bias)
    return layer(x)
