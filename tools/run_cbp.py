"""Launch hk_cbp_fwd / hk_cbp_bwd a few times at C = 512, 14 x 14, D = 6000 (target of rocprofv3 passes).
    python tools/run_cbp.py [B=64] [reps=5] [fwd|bwd|both]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hawkeye_amd import _lib
import hawkeye_amd.functional as F
lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
what = sys.argv[3] if len(sys.argv) > 3 else 'both'
dev = torch.device('cuda:0')
C, HW, D = 512, 196, 6000
P = ctypes.c_void_p
p = lambda t: P(t.data_ptr())
st = P(torch.cuda.current_stream().cuda_stream)
plan = F.CbpPlan(*F.sketch_hashes(C, C, D), D, dev)
x = torch.relu(torch.randn(B, C, HW, device=dev))
y, cr, inv = torch.empty(B, D, device=dev), torch.empty(B, D, device=dev), torch.empty(B, device=dev)
dy, dx = torch.randn(B, D, device=dev), torch.empty_like(x)
nws = lib.hk_cbp_ws_bytes(B, C, HW, D)
ws = torch.empty(nws, dtype=torch.uint8, device=dev)
for _ in range(reps):
    if what in ('fwd', 'both'):
        assert lib.hk_cbp_fwd(p(x), p(plan.blob), p(y), p(cr), p(inv), B, C, HW, D, p(ws), nws, st) == 0
    if what in ('bwd', 'both'):
        assert lib.hk_cbp_bwd(p(x), p(plan.blob), p(y), p(cr), p(inv), p(dy), p(dx), B, C, HW, D, p(ws), nws, st) == 0
torch.cuda.synchronize()
print('ok')
