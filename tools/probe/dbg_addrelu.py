import sys, os
sys.path.insert(0, os.getcwd())
import torch
import hawkeye_amd.functional as HF
from hawkeye_amd.model.backbone.resnet import Bottleneck
blk = Bottleneck(256, 64).cuda().to(memory_format=torch.channels_last).train()
x = torch.randn(4, 256, 14, 14, device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_(True)
calls = []
orig = HF.add_relu_ok
def spy(a, b):
    r = orig(a, b)
    calls.append((r, tuple(a.shape), a.stride(), b.stride(), a.dtype, a.data_ptr() % 16, b.data_ptr() % 16, a.numel() % 4))
    return r
HF.add_relu_ok = spy
y = blk(x)
print(calls)
