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
gk, lo, bi = torch.randn(B, 200, device=dev), torch.randn(B, 200, device=dev), torch.randn(200, device=dev)
for _ in range(reps):
    lib.hk_bcnn_colsum_norm(ptr(x), ptr(cs), ptr(inv), B, C, HW, ptr(wsc), nwsc, stream())
    lib.hk_bcnn_gram_norm(ptr(x), ptr(inv), ptr(y), B, C, HW, stream())
    lib.hk_bcnn_bwd_gemm(ptr(x), ptr(y), ptr(dy), ptr(inv), ptr(dx), ptr(tp), B, C, HW, stream())
    lib.hk_bcnn_bwd_rank1(ptr(dx), ptr(tp), ptr(inv), ptr(cs), B, C, HW, stream())
    # round 5: the one-launch entry points (forward with the column sums in the Gram kernel; backward with t = <y, dy> handed
    # over as a dot product and the rank-1 term in the GEMM kernel's epilogue)
    lib.hk_bcnn_pool_fwd(ptr(x), ptr(y), ptr(inv), ptr(cs), B, C, HW, ptr(wsc), nwsc, stream())
    lib.hk_bcnn_pool_bwd_tdot(ptr(x), ptr(y), ptr(dy), ptr(inv), ptr(cs), ptr(gk), ptr(lo), ptr(bi), 200, ptr(dx), B, C, HW,
                              ptr(wsc), nwsc, stream())
torch.cuda.synchronize()
print('ok', flush=True)

mode = sys.argv[2] if len(sys.argv) > 2 else ''
if mode in ('ns', 'all'):          # the Newton-Schulz chain (grouped nsmm_kernel launches, 256^3 x 64), forward and backward
    import hawkeye_amd.functional as F
    xc = torch.relu(torch.randn(64, 256, 14, 14, device=dev)).requires_grad_(True)
    wt = torch.randn(64, 256, 256, device=dev)
    for _ in range(2):
        xc.grad = None
        # the MPN head as the model runs it: covariance, symmetric-input chain, triuvec - forward and backward
        wv = torch.randn(64, 32896, 1, device=dev)
        (F.triuvec(F.sqrtm(F.covpool(xc), 5, symmetric=True)) * wv).sum().backward()
        xc.grad = None
        (F.sqrtm(F.covpool(xc), 5) * wt).sum().backward()      # the general-input forward (all tiles of every product)
    torch.cuda.synchronize()
    print('ns ok', flush=True)

if mode == 'all':                  # the other heads at the BASELINE config shapes (SURVEY 8a rows A2, A7, A9, A10 + 8f)
    import hawkeye_amd.functional as F
    # A2 compact bilinear pooling, B = 64, C = 512, D = 6000
    plan = F.CbpPlan(*F.sketch_hashes(512, 512, 6000), 6000, dev)
    xb = torch.relu(torch.randn(64, 512, 14, 14, device=dev)).requires_grad_(True)
    wc = torch.randn(64, 6000, device=dev)
    # signed-sqrt Gram (BCNN.py:23-24)
    ws = torch.randn(64, 512 * 512, device=dev)
    # A7 attention pooling (AP-CNN level 3: 16 x 512 x 56 x 56, 200 classes) / A9 ROI refinement / A10 OSME (2 x 16 x 2048 x 7 x 7)
    f3 = torch.randn(16, 512, 56, 56, device=dev, requires_grad=True)            # ROI refinement input (x2: 512 channels)
    fa = torch.randn(16, 256, 56, 56, device=dev, requires_grad=True)            # attention pooling input (FPN level 3: 256 channels)
    a3 = torch.rand(16, 1, 56, 56, device=dev)
    box = torch.tensor([[3.2, 5.9, 40.1, 33.3]] * 8 + [[20.2, 14.9, 43.1, 35.3]] * 8, device=dev)
    drop = torch.tensor([[10., 12., 20., 30.]] * 16, device=dev)
    xo = torch.relu(torch.randn(16, 2048, 7, 7, device=dev)).requires_grad_(True)
    # 8f: classifier (262144 -> 200), n-pairs loss
    yl = torch.randn(64, 262144, device=dev)
    wl, bl = torch.randn(200, 262144, device=dev) * 0.01, torch.zeros(200, device=dev)
    parts = torch.randn(32, 2, 1024, device=dev, requires_grad=True)
    tg = torch.arange(16, device=dev).repeat_interleave(2)
    xb16 = xb.detach()[:16].clone().requires_grad_(True)                          # configs/CBCNN_S2.yaml's batch
    xci = torch.relu(torch.randn(20, 2048, 49, device=dev)).requires_grad_(True)   # CIN channel interaction (CIN.py:24-60) at its config
    wci = torch.randn(20, device=dev, requires_grad=True)
    gci = torch.randn(20, 2048, 49, device=dev)
    gl = torch.randn(64, 200, device=dev)
    dyl, dwl, dbl = torch.empty_like(yl), torch.empty_like(wl), torch.empty_like(bl)
    for _ in range(reps):
        xb.grad = None
        (F.compact_bilinear_pool(xb, plan) * wc).sum().backward()
        xb16.grad = None
        (F.compact_bilinear_pool(xb16, plan) * wc[:16]).sum().backward()
        xci.grad = None
        yci, wsci = F.cin_sci(xci)
        ((yci * gci).sum() + (F.cin_cci(wsci, xci, wci) * gci).sum()).backward()   # SCI + CCI forward and both backward chains
        lib.hk_linear_bwd(ptr(yl), ptr(wl), ptr(gl), ptr(dyl), ptr(dwl), ptr(dbl), 64, 262144, 200, stream())
        xb.grad = None
        (F.bilinear_pool(xb, signed_sqrt=True) * ws).sum().backward()
        fa.grad = None
        g, sg = F.att_pool(fa, a3)
        (g.sum() + sg.sum()).backward()
        f3.grad = None
        F.roi_crop_resize(f3, box, drop, True).sum().backward()
        xo.grad = None
        z = F.osme_gap(xo)
        (F.osme_scale(xo, torch.sigmoid(torch.stack([z, 0.5 * z]))).sum()).backward()      # P = 2 gates [P, N, C]
        F.linear(yl, wl, bl)
        parts.grad = None
        F.npairs_loss(parts, tg).backward()
    torch.cuda.synchronize()
    print('all ok', flush=True)
