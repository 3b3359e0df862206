"""Is MIOpen's fused convolution + bias + ReLU (aten::miopen_convolution_relu) as fast as the plain convolution at the VGG
trunk's shapes (fp32, channels_last)?  If it were, the forward epilogue pass (hk_bias_relu_fwd) could go.
    python tools/probe/conv_relu_probe.py"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hawkeye_amd.miopen_cache import use_in_tree_cache
use_in_tree_cache()
import torch
import torch.nn.functional as F
import hawkeye_amd.functional as HF

dev = torch.device('cuda:0')


def t(fn, it=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


for (n, cin, cout, hw) in ((64, 64, 64, 448), (64, 128, 128, 224), (64, 256, 256, 112), (64, 512, 512, 56), (64, 512, 512, 28)):
    x = torch.randn(n, cin, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 3, 3, device=dev) * 0.02).contiguous(memory_format=torch.channels_last)
    b = torch.randn(cout, device=dev) * 0.1
    with torch.no_grad():
        plain = t(lambda: F.conv2d(x, w, None, 1, 1))
        ours = t(lambda: HF.bias_relu(F.conv2d(x, w, None, 1, 1), b))
        try:
            fused = t(lambda: torch.ops.aten.miopen_convolution_relu(x, w, b, [1, 1], [1, 1], [1, 1], 1))
            y1 = torch.ops.aten.miopen_convolution_relu(x, w, b, [1, 1], [1, 1], [1, 1], 1)
            y2 = HF.bias_relu(F.conv2d(x, w, None, 1, 1), b)
            err = float((y1 - y2).abs().max() / y2.abs().max())
            cl = y1.is_contiguous(memory_format=torch.channels_last)
        except Exception as e:  # noqa: BLE001
            fused, err, cl = None, repr(e)[:200], None
    print(f'{n}x{cin}->{cout}@{hw}: conv {plain:.3f} ms, conv + hk_bias_relu {ours:.3f} ms, miopen_convolution_relu {fused} ms (max diff {err}, channels_last out {cl})', flush=True)
