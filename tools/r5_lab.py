"""Round-5 A/B harness (structure of tools/r4_lab.py: alternating rounds in one process, median / first / min):
    python tools/r5_lab.py [pool] [cov] [cbp] [linear]
pool:  hk_bcnn_pool_fwd / its Gram kernel; hk_bcnn_pool_bwd (two launches) vs hk_bcnn_pool_bwd_tdot (t handed over as a
       dot product: one launch)
cov:   hk_cov_pool_fwd / bwd        cbp: hk_cbp_fwd / bwd at B = 64 and 16        linear: the classifier at BCNN's shape
Prints one JSON object (also gpurun_out/r5_lab_<groups>.json)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch

import r4_lab as L
from r4_lab import dev, lib, p, st

L.DEFAULTS['bwd_fold'] = 0
L.DEFAULTS['fwd_fold'] = 0


def g_pool():
    B, C, HW, K = 64, 512, 196, 200
    x = torch.relu(torch.randn(B, C, HW, device=dev))
    y = torch.empty(B, C * C, device=dev)
    dy = torch.randn(B, C * C, device=dev) * 1e-3
    dx = torch.empty_like(x)
    inv, cs = torch.empty(B, device=dev), torch.empty(B, HW, device=dev)
    g, lo, bi = torch.randn(B, K, device=dev), torch.randn(B, K, device=dev), torch.randn(K, device=dev)
    nws = lib.hk_bcnn_pool_ws_bytes(B, C, HW)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    fl = 2.0 * B * C * C * HW
    by_f, by_b = 4.0 * B * C * HW + 4.0 * B * C * C, 8.0 * B * C * C + 8.0 * B * C * HW
    fwd = lambda: lib.hk_bcnn_pool_fwd(p(x), p(y), p(inv), p(cs), B, C, HW, p(ws), nws, st())
    gram = lambda: lib.hk_bcnn_gram_norm(p(x), p(inv), p(y), B, C, HW, st())
    bwd = lambda: lib.hk_bcnn_pool_bwd(p(x), p(y), p(dy), p(inv), p(cs), p(dx), B, C, HW, p(ws), nws, st())
    gemm = lambda: lib.hk_bcnn_bwd_gemm(p(x), p(y), p(dy), p(inv), p(dx), p(ws), B, C, HW, st())
    tdot = lambda: lib.hk_bcnn_pool_bwd_tdot(p(x), p(y), p(dy), p(inv), p(cs), p(g), p(lo), p(bi), K, p(dx), B, C, HW, p(ws), nws, st())
    fwd()
    items = [('hk_bcnn_pool_fwd (entry point: one launch)', {}, fwd, fl * 36 / 64, by_f),
             ('hk_bcnn_pool_fwd, two launches (fwd_fold=-1)', dict(fwd_fold=-1), fwd, fl * 36 / 64, by_f),
             ('hk_bcnn_gram_norm (Gram kernel alone)', {}, gram, fl * 36 / 64, by_f),
             ('hk_bcnn_pool_bwd, two launches', {}, bwd, fl, by_b),
             ('hk_bcnn_bwd_gemm (GEMM kernel alone, t partials out)', {}, gemm, fl, by_b),
             ('hk_bcnn_pool_bwd_tdot, one launch (t known)', dict(bwd_fold=0), tdot, fl, by_b)]
    return [L.run_group('BCNN pool 64 x 512 x 14 x 14', items)]


def g_cov():
    B, C, HW = 64, 256, 196
    x = torch.relu(torch.randn(B, C, HW, device=dev))
    cov, mu = torch.empty(B, C, C, device=dev), torch.empty(B, C, device=dev)
    gg, dx = torch.randn(B, C, C, device=dev), torch.empty_like(x)
    fl = 2.0 * B * C * C * HW
    fwd = lambda: lib.hk_cov_pool_fwd(p(x), p(cov), p(mu), B, C, HW, st())
    bwd = lambda: lib.hk_cov_pool_bwd(p(x), p(mu), p(gg), p(dx), B, C, HW, st())
    fwd()
    items = [('hk_cov_pool_fwd', {}, fwd, fl * 10 / 16, 4.0 * B * C * HW + 4.0 * B * C * C),
             ('hk_cov_pool_bwd', {}, bwd, fl, 4.0 * B * C * C + 8.0 * B * C * HW)]
    return [L.run_group('covariance 64 x 256 x 14 x 14', items)]


def g_cbp():
    """hk_cbp_fwd / hk_cbp_bwd through the C ABI at B = 64 and the yaml batch 16; with R5_ALT_LIB also the same calls into a
    second build of the library (its own plan: the gather lists are built by hk_cbp_plan_build) in the same rounds."""
    import ctypes
    import hawkeye_amd.functional as F
    from hawkeye_amd import _lib as LL
    out = []
    C, D, HW = 512, 6000, 196
    h1, s1, h2, s2 = F.sketch_hashes(C, C, D)
    libs = [('', lib)]
    alt = os.environ.get('R5_ALT_LIB')
    if alt:
        lib2 = ctypes.CDLL(alt)
        for name, (res, args) in LL.SIGNATURES.items():
            if hasattr(lib2, name):
                getattr(lib2, name).restype, getattr(lib2, name).argtypes = res, args
        libs.append((' of ' + os.path.basename(alt), lib2))
    plans = []
    for tag, lb in libs:
        blob = torch.empty(lb.hk_cbp_plan_bytes(C, D), dtype=torch.uint8, device=dev)
        assert lb.hk_cbp_plan_build(h1.ctypes.data, s1.ctypes.data, h2.ctypes.data, s2.ctypes.data, C, D, p(blob), st()) == 0
        plans.append(blob)
    for B in (64, 16):
        x = torch.relu(torch.randn(B, C, HW, device=dev))
        y, craw, inv = torch.empty(B, D, device=dev), torch.empty(B, D, device=dev), torch.empty(B, device=dev)
        dy, dx = torch.randn(B, D, device=dev), torch.empty_like(x)
        nws = lib.hk_cbp_ws_bytes(B, C, HW, D)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        fl = 2.0 * B * C * C * HW
        items = []
        for (tag, lb), blob in zip(libs, plans):
            items.append(('hk_cbp_fwd' + tag, {}, (lambda lb=lb, blob=blob: lb.hk_cbp_fwd(p(x), p(blob), p(y), p(craw), p(inv), B, C, HW, D, p(ws), nws, st())), fl * 36 / 64, None))
            items.append(('hk_cbp_bwd' + tag, {}, (lambda lb=lb, blob=blob: lb.hk_cbp_bwd(p(x), p(blob), p(y), p(craw), p(inv), p(dy), p(dx), B, C, HW, D, p(ws), nws, st())), fl, None))
        items[0][2]()
        out.append(L.run_group(f'compact bilinear pooling B = {B}', items))
    return out


def g_linear():
    B, J, K = 64, 262144, 200
    y = torch.randn(B, J, device=dev)
    w = torch.randn(K, J, device=dev) * 0.01
    bias, g, o = torch.zeros(K, device=dev), torch.randn(B, K, device=dev), torch.empty(B, K, device=dev)
    dy, dw, db = torch.empty(B, J, device=dev), torch.empty(K, J, device=dev), torch.empty(K, device=dev)
    nws = lib.hk_linear_ws_bytes(B, J, K)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    fl, by = 2.0 * B * J * K, 4.0 * (K * J + B * J + B * K)
    fwd = lambda: lib.hk_linear_fwd(p(y), p(w), p(bias), p(o), B, J, K, p(ws), nws, st())
    bwd = lambda: lib.hk_linear_bwd(p(y), p(w), p(g), p(dy), p(dw), p(db), B, J, K, st())
    items = [('hk_linear_fwd', {}, fwd, fl, by), ('hk_linear_bwd (dy + dW + db)', {}, bwd, 2 * fl, 2 * by)]
    alt = os.environ.get('R5_ALT_LIB')              # a second build of the library (an older kernel) timed in the same rounds
    if alt:
        import ctypes
        from hawkeye_amd import _lib as LL
        lib2 = ctypes.CDLL(alt)
        for name, (res, args) in LL.SIGNATURES.items():
            if hasattr(lib2, name):
                getattr(lib2, name).restype, getattr(lib2, name).argtypes = res, args
        items.append((f'hk_linear_bwd of {os.path.basename(alt)}', {}, lambda: lib2.hk_linear_bwd(p(y), p(w), p(g), p(dy), p(dw), p(db), B, J, K, st()), 2 * fl, 2 * by))
        items.append((f'hk_linear_fwd of {os.path.basename(alt)}', {}, lambda: lib2.hk_linear_fwd(p(y), p(w), p(bias), p(o), B, J, K, p(ws), nws, st()), fl, by))
    return [L.run_group('classifier 64 x 262144 -> 200', items)]


GROUPS = {'pool': g_pool, 'cov': g_cov, 'cbp': g_cbp, 'linear': g_linear}

if __name__ == '__main__':
    want = [a for a in sys.argv[1:] if a in GROUPS] or list(GROUPS)
    res = {'device': torch.cuda.get_device_name(0), 'rounds': L.ROUNDS, 'iters': L.ITERS, 'groups': []}
    for gname in want:
        res['groups'] += GROUPS[gname]()
    txt = json.dumps(res, indent=1)
    print(txt)
    tag = os.environ.get('R5_TAG', '_'.join(want))
    if os.path.isdir(os.path.join(ROOT, 'gpurun_out')):
        with open(os.path.join(ROOT, 'gpurun_out', 'r5_lab_' + tag + '.json'), 'w') as f:
            f.write(txt)
