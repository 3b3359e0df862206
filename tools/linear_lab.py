"""Decomposition of linear_bwd64_kernel at the BCNN shape (64 x 262144 -> 200) with its timing-only instances
(tools/probe/linear_lab.hip): the whole kernel next to the same kernel without MFMAs / without loads / without stores /
without fragment reads, alternating over five rounds.      python tools/linear_lab.py > gpurun_out/linear_lab.json"""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lab = ctypes.CDLL(os.path.join(ROOT, 'tools', 'probe', 'libhk_linear_lab.so'))
P = ctypes.c_void_p
lab.hk_probe_linear_bwd64.argtypes = [ctypes.c_int] + [P] * 6 + [ctypes.c_int] * 4 + [P]
dev = torch.device('cuda:0')
B, J, K = 64, 262144, 200
y, w, g = torch.randn(B, J, device=dev), torch.randn(K, J, device=dev) * 0.01, torch.randn(B, K, device=dev)
dy, dw, db = torch.empty(B, J, device=dev), torch.empty(K, J, device=dev), torch.empty(K, device=dev)
st = lambda: P(torch.cuda.current_stream().cuda_stream)
p = lambda t: P(t.data_ptr()) if t is not None else None
MODES = [(0, 'whole kernel'), (1, 'no MFMAs'), (2, 'no loads in the loop'), (4, 'no stores'), (6, 'no loads, no stores (MFMA + fragment reads + barriers)'),
         (14, 'no loads, no stores, no fragment reads (MFMA + barriers)'), (7, 'barriers + fragment reads only'), (9, 'loads + stores only'),
         (8, 'no fragment reads'), (16, 'default-policy stores instead of nt (results correct)'), (32, 'nt LDS-DMA loads (results correct)'), (48, 'default-policy stores + nt loads')]
ROLES = [('both roles', True, True), ('dy role only', True, False), ('dW role only', False, True)]
res = {}
for rname, wdy, wdw in ROLES:
    out = {m: [] for m, _ in MODES}
    for rnd in range(5):
        for m, _ in MODES:
            fn = lambda: lab.hk_probe_linear_bwd64(m, p(g), p(w), p(y), p(dy) if wdy else None, p(dw) if wdw else None, p(db), B, J, K, 1, st())
            for _ in range(3):
                assert fn() == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            out[m].append(round(e0.elapsed_time(e1) / 20 * 1e3, 1))
    res[rname] = {name: {'us_median': sorted(out[m])[2], 'us_min': min(out[m])} for m, name in MODES}
# ---- forward (linear_skinny_kernel<13, 4>)
lab.hk_probe_linear_fwd.argtypes = [ctypes.c_int, P, P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, P]
part = torch.empty(256 * B * K, device=dev)
FMODES = [(0, 'whole kernel'), (1, 'no MFMAs (LDS-DMA stream + fragment reads + barriers)'), (2, 'no loads in the loop (MFMA + fragment reads + barriers)'),
          (10, 'no loads, no fragment reads (MFMA + barriers)'), (9, 'loads only'), (3, 'fragment reads + barriers only'), (8, 'no fragment reads')]
FM2 = [(m, n, 1) for m, n in FMODES] + [(0, 'whole kernel, contiguous slabs', 0), (9, 'loads only, contiguous slabs', 0), (32, 'nt LDS-DMA loads (results correct)', 1)]
out = {(m, wk): [] for m, _, wk in FM2}
for rnd in range(5):
    for m, _, wk in FM2:
        fn = lambda: lab.hk_probe_linear_fwd(m, p(y), p(w), p(part), B, J, K, wk, st())
        for _ in range(3):
            assert fn() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[(m, wk)].append(round(e0.elapsed_time(e1) / 20 * 1e3, 1))
res['forward'] = {name: {'us_median': sorted(out[(m, wk)])[2], 'us_min': min(out[(m, wk)])} for m, name, wk in FM2}
# the same kernels on ZERO operands (no data toggling: the chip's power management gives the clock back): if the whole kernel
# gains much more than its two sides, the overlap loss is power, not structure
yz, wz = torch.zeros_like(y), torch.zeros_like(w)
outz = {}
for rnd in range(5):
    for m, name in ((0, 'whole kernel'), (9, 'loads only'), (10, 'no loads, no fragment reads (MFMA + barriers)')):
        fn = lambda: lab.hk_probe_linear_fwd(m, p(yz), p(wz), p(part), B, J, K, 1, st())
        for _ in range(3):
            assert fn() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        outz.setdefault(name, []).append(round(e0.elapsed_time(e1) / 20 * 1e3, 1))
res['forward, zero operands'] = {k: {'us_median': sorted(v)[2], 'us_min': min(v)} for k, v in outz.items()}
json.dump(res, sys.stdout, indent=1)
