"""hk_linear_fwd / hk_linear_bwd at the BCNN classifier shape a few times - the target of rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hawkeye_amd import _lib
from hawkeye_amd._lib import ptr, stream
lib = _lib.load()
dev = torch.device('cuda:0')
B, J, K = 64, 262144, 200
y, w, g = torch.randn(B, J, device=dev), torch.randn(K, J, device=dev) * 0.01, torch.randn(B, K, device=dev)
b, out = torch.zeros(K, device=dev), torch.empty(B, K, device=dev)
dy, dw, db = torch.empty_like(y), torch.empty_like(w), torch.empty_like(b)
nws = lib.hk_linear_ws_bytes(B, J, K)
ws = torch.empty(nws, dtype=torch.uint8, device=dev)
for _ in range(12):
    assert lib.hk_linear_fwd(ptr(y), ptr(w), ptr(b), ptr(out), B, J, K, ptr(ws), nws, stream()) == 0
    assert lib.hk_linear_bwd(ptr(y), ptr(w), ptr(g), ptr(dy), ptr(dw), ptr(db), B, J, K, stream()) == 0
torch.cuda.synchronize()
print('ok', flush=True)
