"""A/B of the Gram-backward kernels in the PRODUCT build, alternating over several rounds in one process (clock state and
box-to-box differences move single measurements by several per cent).  B = 64, C = 512, 14 x 14.
    python tools/bwd_ab.py"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from hawkeye_amd import _lib
lib = _lib.load()
P = ctypes.c_void_p
dev = torch.device('cuda:0')
B, C, HW = 64, 512, 196
x = torch.relu(torch.randn(B, C, HW, device=dev))
y, dy, dx = torch.empty(B, C * C, device=dev), torch.randn(B, C * C, device=dev), torch.empty_like(x)
inv, cs, tp = torch.empty(B, device=dev), torch.empty(B, HW, device=dev), torch.empty(B, C // 64, device=dev)
nws = lib.hk_bcnn_pool_ws_bytes(B, C, HW)
ws = torch.empty(nws, dtype=torch.uint8, device=dev)
st = P(torch.cuda.current_stream().cuda_stream)
p = lambda t: P(t.data_ptr())
assert lib.hk_bcnn_colsum_norm(p(x), p(cs), p(inv), B, C, HW, p(ws), nws, st) == 0
assert lib.hk_bcnn_gram_norm(p(x), p(inv), p(y), B, C, HW, st) == 0
names = {(1, 0): '64-row kernel (bwd_v=1)', (5, 0): '128-row, register-staged (bwd_v=5)',
         (9, 0): '128-row, LDS-DMA (bwd_v=9, default)'}
out = {n: [] for n in names.values()}
for rnd in range(5):
    for (v, dg), tag in names.items():
        lib.hk_tuning_set(b'bwd_v', v)
        for _ in range(3):
            assert lib.hk_bcnn_bwd_gemm(p(x), p(y), p(dy), p(inv), p(dx), p(tp), B, C, HW, st) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            lib.hk_bcnn_bwd_gemm(p(x), p(y), p(dy), p(inv), p(dx), p(tp), B, C, HW, st)
        e1.record()
        torch.cuda.synchronize()
        out[tag].append(round(e0.elapsed_time(e1) / 30 * 1e3, 1))
lib.hk_tuning_set(b'bwd_v', 0)
flops = 2.0 * B * C * C * HW
res = {'shape': 'B=64 C=512 14x14', 'us_per_round': out,
       'median_us': {k: sorted(v)[len(v) // 2] for k, v in out.items()},
       'frac_of_157.3_TF': {k: round(flops / (sorted(v)[len(v) // 2] * 1e-6) / 157.3e12, 3) for k, v in out.items()}}
print(json.dumps(res, indent=1))
