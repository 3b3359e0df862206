"""Launch the BCNN pooling-head kernels a few times at the metric's shape (B=64, C=512, 14x14) - the target for
`rocprofv3 --pmc ...` passes (counter collection on the whole training step would take minutes)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hawkeye_amd import _lib
from hawkeye_amd._lib import ptr, stream

B, C, HW = 64, 512, 196
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
lib = _lib.load()
dev = torch.device('cuda:0')
x = torch.relu(torch.randn(B, C, HW, device=dev))
y = torch.empty(B, C * C, device=dev)
dy = torch.randn(B, C * C, device=dev)
dx = torch.empty_like(x)
inv = torch.empty(B, device=dev)
cs = torch.empty(B, HW, device=dev)
tp = torch.empty(B, C // 64, device=dev)
nwsc = lib.hk_bcnn_pool_ws_bytes(B, C, HW)
wsc = torch.empty(nwsc, dtype=torch.uint8, device=dev)
for _ in range(reps):
    lib.hk_bcnn_colsum_norm(ptr(x), ptr(cs), ptr(inv), B, C, HW, ptr(wsc), nwsc, stream())
    lib.hk_bcnn_gram_norm(ptr(x), ptr(inv), ptr(y), B, C, HW, stream())
    lib.hk_bcnn_bwd_gemm(ptr(x), ptr(y), ptr(dy), ptr(inv), ptr(dx), ptr(tp), B, C, HW, stream())
    lib.hk_bcnn_bwd_rank1(ptr(dx), ptr(tp), ptr(inv), ptr(cs), B, C, HW, stream())
torch.cuda.synchronize()
print('ok')

if len(sys.argv) > 2 and sys.argv[2] == 'ns':          # also the Newton-Schulz chain (generic bgemm_kernel, 256^3 x 64)
    import hawkeye_amd.functional as F
    cov = F.covpool(torch.relu(torch.randn(64, 256, 14, 14, device=dev)))
    for _ in range(2):
        F.sqrtm(cov, 5)
    torch.cuda.synchronize()
    print('ns ok')
