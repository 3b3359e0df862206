"""Launch the Fast MPN-COV head at the metric's shape (B = 64, d = 256, 14x14, iterN = 5) a few times: the target for
`rocprofv3 --kernel-trace --stats` and the `--pmc` passes of the Newton-Schulz chain (profiles/r2_ns_*).
    python tools/ns_profile.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hawkeye_amd import _lib
from hawkeye_amd._lib import ptr, stream

B, d, HW = 64, 256, 196
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
lib = _lib.load()
dev = torch.device('cuda:0')
x = torch.relu(torch.randn(B, d, HW, device=dev))
cov, mu = torch.empty(B, d, d, device=dev), torch.empty(B, d, device=dev)
out, na = torch.empty(B, d, d, device=dev), torch.empty(B, device=dev)
ys, zs = torch.empty(B, 4, d, d, device=dev), torch.empty(B, 4, d, d, device=dev)
g, da, dx = torch.randn(B, d, d, device=dev).triu(), torch.empty(B, d, d, device=dev), torch.empty_like(x)
nwf, nwb = lib.hk_ns_sqrtm_ws_bytes(B, d, 5, 0), lib.hk_ns_sqrtm_ws_bytes(B, d, 5, 1)
wf, wb = torch.empty(nwf, dtype=torch.uint8, device=dev), torch.empty(nwb, dtype=torch.uint8, device=dev)
for _ in range(reps):
    lib.hk_cov_pool_fwd(ptr(x), ptr(cov), ptr(mu), B, d, HW, stream())
    lib.hk_ns_sqrtm_fwd_sym(ptr(cov), ptr(out), ptr(na), ptr(ys), ptr(zs), B, d, 5, ptr(wf), nwf, stream())   # the MPN head's call
    lib.hk_ns_sqrtm_bwd(ptr(cov), ptr(out), ptr(na), ptr(ys), ptr(zs), ptr(g), ptr(da), B, d, 5, ptr(wb), nwb, stream())
    lib.hk_cov_pool_bwd(ptr(x), ptr(mu), ptr(da), ptr(dx), B, d, HW, stream())
torch.cuda.synchronize()
print('ok', float(out.abs().sum()), float(da.abs().sum()))
