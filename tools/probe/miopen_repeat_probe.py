"""Run-to-run repeatability of the PLAIN VGG stack's gradients on this box (torch ops + MIOpen only - none of this repo's kernels):
    python tools/probe/miopen_repeat_probe.py
Found while de-flaking test_vgg_trunk_with_fused_epilogues_equals_the_plain_stack: about one run in eight, the input gradient and
the first convolution's weight gradient of a 4 x 3 x 96 x 64 batch come back 1.2e-2 away from every other run (and from a float64
CPU evaluation) - in the fused stack and in the plain one alike."""
import copy
import os
import sys
sys.path.insert(0, os.getcwd())
import torch
from hawkeye_amd.miopen_cache import use_in_tree_cache
use_in_tree_cache()
from hawkeye_amd.model.backbone import vgg16


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


torch.manual_seed(3)
net = torch.nn.Sequential(*vgg16(pretrained=False).features.children()).cuda().to(memory_format=torch.channels_last)
for shape in ((4, 3, 96, 64), (2, 3, 224, 224), (8, 3, 448, 448)):
    x = torch.randn(*shape, device='cuda').contiguous(memory_format=torch.channels_last)
    first, worst, bad = None, 0.0, 0
    per_layer_bad = {}
    for i in range(16):
        xi = x.clone().requires_grad_(True)
        for p in net.parameters():
            p.grad = None
        y = net(xi)
        y.square().mean().backward()
        torch.cuda.synchronize()
        g = [xi.grad.clone()] + [p.grad.clone() for p in net.parameters()]
        if first is None:
            first = g
            continue
        d = [rel(a, b) for a, b in zip(g, first)]
        worst = max(worst, max(d))
        if max(d) > 1e-4:
            bad += 1
            for k, v in enumerate(d):
                if v > 1e-4:
                    per_layer_bad[k] = per_layer_bad.get(k, 0) + 1
    print(f'{shape}: 15 repeats, worst distance from the first run {worst:.2e}, runs beyond 1e-4: {bad}, by tensor (0 = input, 1.. = parameters in order): {per_layer_bad}', flush=True)
