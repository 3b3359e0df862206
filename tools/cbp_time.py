"""HIP-event timing of hk_cbp_fwd / hk_cbp_bwd only (C=512, 14x14, D=6000).  HK_CBP_BIN=0|1|2 forces a binning kernel (row-sketch, CSR gather, row-scatter)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hawkeye_amd.functional as F
from hawkeye_amd import _lib
from hawkeye_amd._lib import ptr, stream

lib = _lib.load()
dev = torch.device('cuda:0')
C, HW, D = 512, 196, 6000
plan = F.CbpPlan(*F.sketch_hashes(C, C, D), D, dev)
for bb in (16, 64):
    xc = torch.relu(torch.randn(bb, C, HW, device=dev))
    yc = torch.empty(bb, D, device=dev); craw = torch.empty(bb, D, device=dev); invc = torch.empty(bb, device=dev)
    nws = lib.hk_cbp_ws_bytes(bb, C, HW, D); ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    fn = lambda: lib.hk_cbp_fwd(ptr(xc), ptr(plan.blob), ptr(yc), ptr(craw), ptr(invc), bb, C, HW, D, ptr(ws), nws, stream())
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(30):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f'cbp fwd B={bb} cbp_bin={os.environ.get("HK_CBP_BIN", "-1")}: {e0.elapsed_time(e1) / 30 * 1e3:.1f} us', flush=True)
