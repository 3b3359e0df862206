"""Launch the classifier kernels a few times at a plugin's shape - the target for tools/r4_prof.sh (rocprofv3 kernel
trace + PMC passes).      python tools/run_linear.py [bcnn|mpn|osme] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hawkeye_amd import _lib
from hawkeye_amd._lib import ptr, stream

shape = {'bcnn': (64, 262144, 200), 'mpn': (64, 32896, 200), 'osme': (10, 100352, 1024)}[sys.argv[1] if len(sys.argv) > 1 else 'bcnn']
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B, J, K = shape
lib = _lib.load()
dev = torch.device('cuda:0')
y, w, bias = torch.randn(B, J, device=dev), torch.randn(K, J, device=dev) * 0.01, torch.zeros(K, device=dev)
g, o = torch.randn(B, K, device=dev), torch.empty(B, K, device=dev)
dy, dw, db = torch.empty(B, J, device=dev), torch.empty(K, J, device=dev), torch.empty(K, device=dev)
nws = lib.hk_linear_ws_bytes(B, J, K)
ws = torch.empty(nws, dtype=torch.uint8, device=dev)
for _ in range(reps):
    assert lib.hk_linear_fwd(ptr(y), ptr(w), ptr(bias), ptr(o), B, J, K, ptr(ws), nws, stream()) == 0
    assert lib.hk_linear_bwd(ptr(y), ptr(w), ptr(g), ptr(dy), ptr(dw), ptr(db), B, J, K, stream()) == 0
torch.cuda.synchronize()
print('ok', flush=True)
