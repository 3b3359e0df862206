"""Offline-safe ImageNet weights.  The reference downloads them through torch.hub
(model/backbone/vgg.py:83-85, resnet.py:262-266); build/GPU boxes have no network,
so weights are looked up as local files and a missing file falls back to the
reference's random initialisation with one warning.

    HAWKEYE_PRETRAINED_DIR=/path/with/vgg16-397923af.pth,resnet50-0676ba61.pth,...
"""
import os
import warnings

import torch

FILES = {
    'vgg16': 'vgg16-397923af.pth',
    'resnet50': 'resnet50-0676ba61.pth',
    'resnet101': 'resnet101-63fe2227.pth',
}
_warned = set()


def find(arch):
    dirs = [os.environ.get('HAWKEYE_PRETRAINED_DIR', ''),
            os.path.join(torch.hub.get_dir(), 'checkpoints')]
    for d in dirs:
        p = os.path.join(d, FILES[arch]) if d else ''
        if p and os.path.isfile(p):
            return p
    return None


def load(arch):
    """state_dict or None (with a one-time warning)."""
    p = find(arch)
    if p is None:
        if arch not in _warned:
            _warned.add(arch)
            warnings.warn(f'pretrained {arch} weights ({FILES[arch]}) not found locally and there is no network: '
                          f'using random init (set HAWKEYE_PRETRAINED_DIR to load them)')
        return None
    return torch.load(p, map_location='cpu')
