"""Round-6 A/B harness (structure of tools/r4_lab.py: alternating rounds in one process, median / first / min):
    python tools/r6_lab.py [ns] [cov] [lin_mpn] [lin_cbcnn] [cbp] [linear] [pool]
ns:        the Newton-Schulz chain at B = 64, d = 256, iterN = 5: tile width x symmetric schedule x queues
cov:       hk_cov_pool_fwd / bwd                      lin_mpn:   the classifier at MPN's width (64 x 32896 -> 200), slab counts
lin_cbcnn: the classifier at CBCNN's width (B x 6000 -> 200, B = 16 / 64)
cbp / linear / pool: tools/r5_lab.py's groups (R6_ALT_LIB = a second build of the library timed in the same rounds)
Prints one JSON object (also gpurun_out/r6_lab_<tag>.json)."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
if os.environ.get('R6_ALT_LIB'):
    os.environ['R5_ALT_LIB'] = os.environ['R6_ALT_LIB']
import torch

import r4_lab as L
import r5_lab as L5
from r4_lab import dev, lib, p, st


def alt_lib():
    alt = os.environ.get('R6_ALT_LIB')
    if not alt:
        return None
    from hawkeye_amd import _lib as LL
    lib2 = ctypes.CDLL(alt)
    for name, (res, args) in LL.SIGNATURES.items():
        if hasattr(lib2, name):
            getattr(lib2, name).restype, getattr(lib2, name).argtypes = res, args
    return lib2


def g_ns():
    B, d, it = 64, 256, 5
    n = d * d
    x = torch.relu(torch.randn(B, d, 196, device=dev))
    cov, mu = torch.empty(B, d, d, device=dev), torch.empty(B, d, device=dev)
    assert lib.hk_cov_pool_fwd(p(x), p(cov), p(mu), B, d, 196, st()) == 0
    out, na = torch.empty(B, d, d, device=dev), torch.empty(B, device=dev)
    ys, zs = torch.empty(B, it - 1, d, d, device=dev), torch.empty(B, it - 1, d, d, device=dev)
    tv = torch.empty(B, d * (d + 1) // 2, device=dev)
    dout, da = torch.randn(B, d, d, device=dev), torch.empty(B, d, d, device=dev)
    nwf, nwb = lib.hk_ns_sqrtm_ws_bytes(B, d, it, 0), lib.hk_ns_sqrtm_ws_bytes(B, d, it, 1)
    ws = torch.empty(max(nwf, nwb), dtype=torch.uint8, device=dev)
    fwd_sym = lambda: lib.hk_ns_sqrtm_fwd_sym(p(cov), p(out), p(na), p(ys), p(zs), B, d, it, p(ws), nwf, st())
    fwd_tri = lambda: lib.hk_ns_sqrtm_triu_fwd(p(cov), p(out), p(tv), p(na), p(ys), p(zs), B, d, it, 1, p(ws), nwf, st())
    fwd_gen = lambda: lib.hk_ns_sqrtm_fwd(p(cov), p(out), p(na), p(ys), p(zs), B, d, it, p(ws), nwf, st())
    bwd = lambda: lib.hk_ns_sqrtm_bwd(p(cov), p(out), p(na), p(ys), p(zs), p(dout), p(da), B, d, it, p(ws), nwb, st())
    fwd_sym()
    ff, fb = 12 * 2.0 * d ** 3 * B, 38 * 2.0 * d ** 3 * B
    items = [('fwd sym (default)', {}, fwd_sym, ff * 0.75, None), ('fwd sym + triu (what MPN runs)', {}, fwd_tri, ff * 0.75, None)]
    for tn in (64, 128):
        for s in (0, 1, 2, 3):
            items.append((f'fwd sym tn={tn} queues={s + 1}', dict(ns_tn=tn, ns_streams=s), fwd_sym, ff * 0.75, None))
    items += [('fwd general (default)', {}, fwd_gen, ff, None), ('fwd sym off (ns_sym=0)', dict(ns_sym=0), fwd_sym, ff, None),
              ('bwd (default)', {}, bwd, fb * 34 / 38, None)]
    for tn in (64, 128):
        for s in (0, 1, 2, 3):
            items.append((f'bwd tn={tn} queues={s + 1}', dict(ns_tn=tn, ns_streams=s), bwd, fb * 34 / 38, None))
    return [L.run_group('Newton-Schulz chain 64 x 256 x 256, iterN 5', items)]


def _lin_group(name, B, J, K, slab_list):
    y = torch.randn(B, J, device=dev)
    w = torch.randn(K, J, device=dev) * 0.01
    bias, g, o = torch.zeros(K, device=dev), torch.randn(B, K, device=dev), torch.empty(B, K, device=dev)
    dy, dw, db = torch.empty(B, J, device=dev), torch.empty(K, J, device=dev), torch.empty(K, device=dev)
    fl, by = 2.0 * B * J * K, 4.0 * (K * J + B * J + B * K)
    big = max(lib.hk_linear_ws_bytes(B, J, K), 1024 * B * K * 4 + 4096)      # room for any forced slab count
    ws = torch.empty(big, dtype=torch.uint8, device=dev)
    fwd = lambda lb=lib: lb.hk_linear_fwd(p(y), p(w), p(bias), p(o), B, J, K, p(ws), big, st())
    bwd = lambda lb=lib: lb.hk_linear_bwd(p(y), p(w), p(g), p(dy), p(dw), p(db), B, J, K, st())

    def lib_fwd():
        torch.addmm(bias, y, w.t(), out=o)

    def lib_bwd():
        torch.mm(g, w, out=dy)
        torch.mm(g.t(), y, out=dw)
        torch.sum(g, 0, out=db)
    items = [('hk_linear_fwd', {}, fwd, fl, by)]
    items += [(f'hk_linear_fwd linear_slabs={s}', dict(linear_slabs=s), fwd, fl, by) for s in slab_list]
    items += [('rocBLAS fwd (torch.addmm)', {}, lib_fwd, fl, by), ('hk_linear_bwd (dy + dW + db)', {}, bwd, 2 * fl, 2 * by),
              ('hk_linear_bwd generic tiles (linear_slabs=-1)', dict(linear_slabs=-1), bwd, 2 * fl, 2 * by),
              ('rocBLAS bwd (torch.mm x 2 + sum)', {}, lib_bwd, 2 * fl, 2 * by)]
    l2 = alt_lib()
    if l2 is not None:
        tag = ' of ' + os.path.basename(os.environ['R6_ALT_LIB'])
        items += [('hk_linear_fwd' + tag, {}, lambda: fwd(l2), fl, by), ('hk_linear_bwd' + tag, {}, lambda: bwd(l2), 2 * fl, 2 * by)]
    return L.run_group('classifier ' + name, items)


def g_lin_mpn():
    return [_lin_group('MPN 64 x 32896 -> 200', 64, 32896, 200, (-1, 64, 128, 257))]


def g_lin_mpn_sweep():
    return [_lin_group('MPN 64 x 32896 -> 200', 64, 32896, 200, (86, 103, 115, 129, 147, 172, 206)),
            _lin_group('MPN yaml batch 8 x 32896 -> 200', 8, 32896, 200, (64, 103, 129, 172)),
            _lin_group('CBCNN-sized 64 x 8192 -> 200', 64, 8192, 200, (32, 64, 128, 256))]


def g_lin_cbcnn():
    return [_lin_group('CBCNN 16 x 6000 -> 200', 16, 6000, 200, (-1,)), _lin_group('CBCNN 64 x 6000 -> 200', 64, 6000, 200, (-1,)),
            _lin_group('no tail: 16 x 5952 -> 200', 16, 5952, 200, (-1,)), _lin_group('64 x 16384 -> 200', 64, 16384, 200, (-1,))]


GROUPS = {'lin_mpn_sweep': g_lin_mpn_sweep, 'ns': g_ns, 'cov': L5.g_cov, 'lin_mpn': g_lin_mpn, 'lin_cbcnn': g_lin_cbcnn, 'cbp': L5.g_cbp, 'linear': L5.g_linear,
          'pool': L5.g_pool}

if __name__ == '__main__':
    want = [a for a in sys.argv[1:] if a in GROUPS] or ['ns', 'cov', 'lin_mpn', 'lin_cbcnn']
    res = {'device': torch.cuda.get_device_name(0), 'rounds': L.ROUNDS, 'iters': L.ITERS, 'groups': []}
    for gname in want:
        res['groups'] += GROUPS[gname]()
    txt = json.dumps(res, indent=1)
    print(txt)
    tag = os.environ.get('R6_TAG', '_'.join(want))
    if os.path.isdir(os.path.join(ROOT, 'gpurun_out')):
        with open(os.path.join(ROOT, 'gpurun_out', 'r6_lab_' + tag + '.json'), 'w') as f:
            f.write(txt)
