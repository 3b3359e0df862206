"""CIN's channel-interaction module (/root/reference/model/methods/CIN.py:24-60) at the config's shape (B = 20,
C = 2048, 7 x 7 maps) and at a 448^2 input's 14 x 14 maps: hk_cin_sci_fwd / bwd and hk_cin_cci_fwd / bwd by HIP events, next to
the library chain a PyTorch user would run for the same lines (torch.bmm + softmax + torch.bmm = rocBLAS + a softmax kernel) -
a yardstick only; nothing in the package calls it.

    python tools/cin_rows.py [--hw 49,196]      # prints one JSON list
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hawkeye_amd import functional as F  # noqa: E402

dev = torch.device('cuda:0')
rows = []


def timed(fn, iters=10):
    return bench.time_events(fn, iters, rounds=3)[0] * 1e3


def module_rows(B, C, HW):
    x = torch.randn(B, C, HW, device=dev) * 0.5
    g = torch.randn(B, C, HW, device=dev)
    wt = torch.rand(B, device=dev)
    fl = 2.0 * B * C * C * HW
    xg = x.clone().requires_grad_(True)

    def ours_fwd():
        with torch.no_grad():
            return F.cin_sci(x)

    def lib_fwd():
        with torch.no_grad():
            w = torch.softmax(-torch.bmm(x, x.transpose(1, 2)) / HW, dim=-1)
            return torch.bmm(w, x), w

    y, w = ours_fwd()
    yl, wl = lib_fwd()
    err = float((y - yl).norm() / yl.norm())
    us = timed(ours_fwd)
    rows.append({'shape': [B, C, HW], 'row': 'SCI forward (hk_cin_sci_fwd)', 'us': round(us, 1), 'tflops_algorithmic': round(2 * fl / us * 1e-6, 1),
                 'vs_library_rel': err})
    us = timed(lib_fwd)
    rows.append({'shape': [B, C, HW], 'row': 'SCI forward, library chain (bmm + softmax + bmm)', 'us': round(us, 1),
                 'tflops_algorithmic': round(2 * fl / us * 1e-6, 1)})

    def ours_fb():
        xg.grad = None
        y_, w_ = F.cin_sci(xg)
        (y_ * g).sum().backward()

    def lib_fb():
        xg.grad = None
        w_ = torch.softmax(-torch.bmm(xg, xg.transpose(1, 2)) / HW, dim=-1)
        (torch.bmm(w_, xg) * g).sum().backward()

    rows.append({'shape': [B, C, HW], 'row': 'SCI forward + backward (autograd node over hk_cin_sci_fwd / bwd)', 'us': round(timed(ours_fb, 6), 1)})
    rows.append({'shape': [B, C, HW], 'row': 'SCI forward + backward, library chain', 'us': round(timed(lib_fb, 6), 1)})

    def ours_full():
        xg.grad = None
        y_, w_ = F.cin_sci(xg)
        ((y_ * g).sum() + (F.cin_cci(w_, xg, wt) * g).sum()).backward()

    rows.append({'shape': [B, C, HW], 'row': 'SCI + CCI forward and backward (the whole module)', 'us': round(timed(ours_full, 6), 1)})


if __name__ == '__main__':
    hws = [49, 196]
    for a in sys.argv[1:]:
        if a.startswith('--hw'):
            hws = [int(v) for v in sys.argv[sys.argv.index(a) + 1].split(',')]
    for hw in hws:
        try:
            module_rows(20, 2048, hw)
        except Exception as e:  # noqa: BLE001
            rows.append({'shape': [20, 2048, hw], 'error': repr(e)[:300]})
    print(json.dumps(rows), flush=True)
