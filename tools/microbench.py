"""HIP-event timing of every entry point of the C ABI at the BASELINE config shapes, called directly (no autograd,
pre-allocated buffers): achieved algorithmic TFLOP/s / GB/s per op.  Ops that are chains of kernels (Newton-Schulz:
14 / 44 launches) are reported per call.
    python tools/microbench.py [--json out.json]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hawkeye_amd.functional as F
from hawkeye_amd import _lib
from hawkeye_amd._lib import ptr, stream

lib = _lib.load()
dev = torch.device('cuda:0')
PEAK_TF, PEAK_GBS = 157.3, 8000.0
rows = []


def timeit(fn, iters=30, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def report(name, fn, flops=0.0, bytes_=0.0):
    us = timeit(fn)
    tf, gbs = flops / us / 1e6, bytes_ / us / 1e3
    rows.append(dict(op=name, us=round(us, 1), tflops=round(tf, 1), gbs=round(gbs, 0),
                     frac_mfma=round(tf / PEAK_TF, 3), frac_hbm=round(gbs / PEAK_GBS, 3)))
    print(f'{name:40s} {us:9.1f} us  {tf:7.1f} TF/s ({tf / PEAK_TF:5.1%})  {gbs:7.0f} GB/s ({gbs / PEAK_GBS:5.1%})', flush=True)


def E(*shape, dtype=torch.float32):
    return torch.empty(*shape, dtype=dtype, device=dev)


def R(*shape):
    return torch.randn(*shape, device=dev)


B = 64
# ---------------------------------------------------------------- BCNN (C=512, 14x14)
C, HW = 512, 196
x = torch.relu(R(B, C, HW)); y = E(B, C * C); dy = R(B, C * C); dx = E(B, C, HW)
inv = E(B); cs = E(B, HW); tp = E(B, C // 64)
nwsc = lib.hk_bcnn_pool_ws_bytes(B, C, HW); wsc = E(nwsc, dtype=torch.uint8)
fl = 2.0 * B * C * C * HW
report('bcnn colsum_norm', lambda: lib.hk_bcnn_colsum_norm(ptr(x), ptr(cs), ptr(inv), B, C, HW, ptr(wsc), nwsc, stream()), 0, 4.0 * B * C * HW)
report('bcnn gram_norm', lambda: lib.hk_bcnn_gram_norm(ptr(x), ptr(inv), ptr(y), B, C, HW, stream()), fl, 4.0 * B * (C * HW + C * C))
report('bcnn bwd_gemm', lambda: lib.hk_bcnn_bwd_gemm(ptr(x), ptr(y), ptr(dy), ptr(inv), ptr(dx), ptr(tp), B, C, HW, stream()), fl, 8.0 * B * (C * C + C * HW))
report('bcnn bwd_rank1', lambda: lib.hk_bcnn_bwd_rank1(ptr(dx), ptr(tp), ptr(inv), ptr(cs), B, C, HW, stream()), 0, 8.0 * B * C * HW)
# ---------------------------------------------------------------- CBP (C=512, D=6000)
D = 6000
plan = F.CbpPlan(*F.sketch_hashes(C, C, D), D, dev)
for bb in (16, 64):
    xc = torch.relu(R(bb, C, HW)); yc = E(bb, D); craw = E(bb, D); invc = E(bb); dyc = R(bb, D); dxc = E(bb, C, HW)
    nws = lib.hk_cbp_ws_bytes(bb, C, HW, D); ws = E(nws, dtype=torch.uint8)
    flc = 2.0 * bb * C * C * HW
    report(f'cbp fwd B={bb}', lambda: lib.hk_cbp_fwd(ptr(xc), ptr(plan.blob), ptr(yc), ptr(craw), ptr(invc), bb, C, HW, D, ptr(ws), nws, stream()), flc, 4.0 * bb * (C * HW + D))
    report(f'cbp bwd B={bb}', lambda: lib.hk_cbp_bwd(ptr(xc), ptr(plan.blob), ptr(yc), ptr(craw), ptr(invc), ptr(dyc), ptr(dxc), bb, C, HW, D, ptr(ws), nws, stream()), flc, 4.0 * bb * (2 * C * HW + 2 * D))
# ---------------------------------------------------------------- MPN-COV (C=256, 14x14, iterN=5)
d = 256
xm = torch.relu(R(B, d, HW)); cov = E(B, d, d); mu = E(B, d); g = R(B, d, d); dxm = E(B, d, HW)
report('cov_pool fwd', lambda: lib.hk_cov_pool_fwd(ptr(xm), ptr(cov), ptr(mu), B, d, HW, stream()), 2.0 * B * d * d * HW, 4.0 * B * (d * HW + d * d))
report('cov_pool bwd', lambda: lib.hk_cov_pool_bwd(ptr(xm), ptr(mu), ptr(g), ptr(dxm), B, d, HW, stream()), 2.0 * B * d * d * HW, 4.0 * B * (2 * d * HW + d * d))
out = E(B, d, d); na = E(B); ys = E(B, 4, d, d); zs = E(B, 4, d, d); da = E(B, d, d)
nwf = lib.hk_ns_sqrtm_ws_bytes(B, d, 5, 0); nwb = lib.hk_ns_sqrtm_ws_bytes(B, d, 5, 1)
wsf = E(nwf, dtype=torch.uint8); wsb = E(nwb, dtype=torch.uint8)
report('ns_sqrtm fwd (12 GEMM 256^3)', lambda: lib.hk_ns_sqrtm_fwd(ptr(cov), ptr(out), ptr(na), ptr(ys), ptr(zs), B, d, 5, ptr(wsf), nwf, stream()), 12 * 2.0 * B * d ** 3, 4.0 * B * d * d * 10)
report('ns_sqrtm bwd (38 GEMM 256^3)', lambda: lib.hk_ns_sqrtm_bwd(ptr(cov), ptr(out), ptr(na), ptr(ys), ptr(zs), ptr(g), ptr(da), B, d, 5, ptr(wsb), nwb, stream()), 38 * 2.0 * B * d ** 3, 4.0 * B * d * d * 12)
tv = E(B, d * (d + 1) // 2)
report('triu_vec fwd', lambda: lib.hk_triu_vec_fwd(ptr(out), ptr(tv), B, d, stream()), 0, 4.0 * B * 32896 * 2)
report('triu_vec bwd', lambda: lib.hk_triu_vec_bwd(ptr(tv), ptr(da), B, d, stream()), 0, 4.0 * B * (32896 + 65536))
# ---------------------------------------------------------------- AP-CNN (B=16 per GPU in the yaml)
ba = 16
for hw in (56, 28, 14):
    n = hw * hw
    ff = R(ba, 256, n); aa = torch.rand(ba, n, device=dev); gp = E(ba, 256); sg = E(ba, 256)
    dgp = R(ba, 256); dsg = R(ba, 256); dff = E(ba, 256, n); daa = E(ba, n)
    report(f'att_pool fwd {hw}x{hw} B={ba}', lambda: lib.hk_att_pool_fwd(ptr(ff), ptr(aa), ptr(gp), ptr(sg), ba, 256, n, stream()), 0, 4.0 * ba * 256 * n)
    report(f'att_pool bwd {hw}x{hw} B={ba}', lambda: lib.hk_att_pool_bwd(ptr(ff), ptr(aa), ptr(dgp), ptr(dsg), ptr(dff), ptr(daa), ba, 256, n, stream()), 0, 8.0 * ba * 256 * n)
masks = [torch.rand(ba, s * s, device=dev) for s in (56, 28, 14)]
lv = ((56, 8, 64., 5, 11, 44), (28, 16, 128., 3, 5, 22), (14, 32, 256., 1, 2, 11))
tabs = [(E(ba, k, 5), E(ba, dtype=torch.int32)) for (_, _, _, k, _, _) in lv]


def roi_all():
    for m, (hw, s, a, k, r0, r1), (rt, ct) in zip(masks, lv, tabs):
        lib.hk_att_roi_select(ptr(m), ptr(rt), ptr(ct), ba, hw, hw, s, a, 448, 448, r0, r1, r0, r1, 0.05, k, stream())


report('att_roi_select x3 levels', roi_all)
u = torch.rand(ba, 2, device=dev)
box, drop = F.roi_boxes(tabs, u, 8.0)
x2 = R(ba, 512, 56, 56); y2 = E(ba, 512, 56, 56)
report(f'roi_crop_resize fwd B={ba}', lambda: lib.hk_roi_crop_resize_fwd(ptr(x2), ptr(box), ptr(drop), ptr(y2), ba, 512, 56, 56, 1, stream()), 0, 8.0 * ba * 512 * 3136)
report(f'roi_crop_resize bwd B={ba}', lambda: lib.hk_roi_crop_resize_bwd(ptr(y2), ptr(box), ptr(drop), ptr(x2), ba, 512, 56, 56, 1, stream()), 0, 8.0 * ba * 512 * 3136)
# ---------------------------------------------------------------- OSME (N=10, C=2048, 7x7, P=2)
xo = R(10, 2048, 49); mo = torch.rand(2, 10, 2048, device=dev); so = E(2, 10, 2048, 49); zo = E(10, 2048)
dxo = E(10, 2048, 49); dmo = E(2, 10, 2048)
report('osme_gap N=10', lambda: lib.hk_osme_gap(ptr(xo), ptr(zo), 10, 2048, 49, stream()), 0, 4.0 * 10 * 2048 * 49)
report('osme_scale fwd N=10 P=2', lambda: lib.hk_osme_scale_fwd(ptr(xo), ptr(mo), ptr(so), 2, 10, 2048, 49, stream()), 0, 4.0 * 10 * 2048 * 49 * 3)
report('osme_scale bwd N=10 P=2', lambda: lib.hk_osme_scale_bwd(ptr(xo), ptr(mo), ptr(so), None, ptr(dxo), ptr(dmo), 2, 10, 2048, 49, stream()), 0, 4.0 * 10 * 2048 * 49 * 4)
# ---------------------------------------------------------------- SURVEY 8f rows (first timed in round 2)
Bc, J, K = 64, 262144, 200
yl, wl, bl, gl = R(Bc, J), R(K, J) * 0.01, R(K), R(Bc, K)
ol, dyl, dwl, dbl = E(Bc, K), E(Bc, J), E(K, J), E(K)
nwl = lib.hk_linear_ws_bytes(Bc, J, K); wsl = E(nwl, dtype=torch.uint8)
report('linear fwd 262144->200 B=64', lambda: lib.hk_linear_fwd(ptr(yl), ptr(wl), ptr(bl), ptr(ol), Bc, J, K, ptr(wsl), nwl, stream()),
       2.0 * Bc * J * K, 4.0 * (Bc * J + K * J))
report('linear bwd 262144->200 B=64', lambda: lib.hk_linear_bwd(ptr(yl), ptr(wl), ptr(gl), ptr(dyl), ptr(dwl), ptr(dbl), Bc, J, K, stream()),
       4.0 * Bc * J * K, 4.0 * (2 * Bc * J + 2 * K * J))
del yl, wl, dyl, dwl
xp_ = R(10, 2, 1024); lab = (torch.arange(10, device=dev) // 2).to(torch.int32); ls, dxp = E(1), E(10, 2, 1024)
nwn = lib.hk_npairs_ws_bytes(20, 1024); wsn = E(nwn, dtype=torch.uint8)
report('npairs loss+grad b=10 p=2', lambda: lib.hk_npairs_loss(ptr(xp_), ptr(lab), ptr(ls), ptr(dxp), 10, 2, 1024, ptr(wsn), nwn, stream()))
Bi, Ci, HWi = 20, 2048, 49
xi = torch.relu(R(Bi, Ci, HWi)); wi, yi = E(Bi, Ci, Ci), E(Bi, Ci, HWi)
report('cin sci fwd B=20 C=2048', lambda: lib.hk_cin_sci_fwd(ptr(xi), ptr(wi), ptr(yi), Bi, Ci, HWi, stream()), 4.0 * Bi * Ci * Ci * HWi,
       4.0 * Bi * 2 * Ci * Ci)
u8 = torch.randint(0, 256, (64, 448, 448, 3), dtype=torch.uint8, device=dev); oi = E(64, 3, 448, 448)
import ctypes
m3 = (ctypes.c_float * 3)(0.485, 0.456, 0.406); s3 = (ctypes.c_float * 3)(0.229, 0.224, 0.225)
report('image_finalize B=64 448x448', lambda: lib.hk_image_finalize(ptr(u8), ctypes.cast(m3, ctypes.c_void_p), ctypes.cast(s3, ctypes.c_void_p),
                                                                     None, ptr(oi), 64, 448, 448, 0, stream()), 0, 15.0 * 64 * 448 * 448)
if '--json' in sys.argv:
    json.dump(rows, open(sys.argv[sys.argv.index('--json') + 1], 'w'), indent=1)
