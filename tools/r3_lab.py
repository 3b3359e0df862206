"""Round-3 A/B harness: HIP-event timings of head kernels and their variants, ALTERNATING over several rounds in one
process (clock state moves single measurements by several per cent; the first round of a process is cold) - median and
first-round value per item.  Direct C-ABI calls on torch's current stream.
    python tools/r3_lab.py [group ...]        # groups: bcnn cov ssqrt cbp ns linear small ; default: all
Prints one JSON object (also written to gpurun_out/r3_lab.json when that directory exists)."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from hawkeye_amd import _lib
import hawkeye_amd.functional as F

lib = _lib.load()
P = ctypes.c_void_p
dev = torch.device('cuda:0')
st = lambda: P(torch.cuda.current_stream().cuda_stream)
p = lambda t: P(t.data_ptr()) if t is not None else None
ROUNDS, ITERS = 5, 20
PEAK_TF, PEAK_GBS = 157.3, 8000.0


def knobs(**kw):
    for k, v in kw.items():
        assert lib.hk_tuning_set(k.encode(), int(v)) == 0, k


DEFAULTS = dict(bwd_v=0, cbp_bin=-1, ns_streams=1, ns_tn=0, linear_slabs=0, bcnn_generic=0, ns_sym=1, sched_b=0, ns_flow=0)


def run_group(title, items, flops=None, bytes_=None):
    """items: list of (tag, knob dict, fn) - fn() enqueues one call and returns its rc."""
    out = {tag: [] for tag, _, _ in items}
    for rnd in range(ROUNDS):
        for tag, kn, fn in items:
            knobs(**DEFAULTS)
            knobs(**kn)
            for _ in range(3):
                rc = fn()
                assert rc in (0, None), (tag, rc)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(ITERS):
                fn()
            e1.record()
            torch.cuda.synchronize()
            out[tag].append(round(e0.elapsed_time(e1) / ITERS * 1e3, 2))
    knobs(**DEFAULTS)
    res = {}
    for tag, v in out.items():
        med = sorted(v)[len(v) // 2]
        r = {'us': med, 'us_first_round': v[0], 'us_min': min(v)}
        if flops:
            r['tflops'] = round(flops / med / 1e6, 1)
            r['frac_mfma'] = round(flops / med / 1e6 / PEAK_TF, 3)
        if bytes_:
            r['gbs'] = round(bytes_ / med / 1e3, 0)
            r['frac_hbm'] = round(bytes_ / med / 1e3 / PEAK_GBS, 3)
        res[tag] = r
    return {'group': title, 'rows': res}


def g_bcnn():
    B, C, HW = 64, 512, 196
    x = torch.relu(torch.randn(B, C, HW, device=dev))
    y, dy, dx = torch.empty(B, C * C, device=dev), torch.randn(B, C * C, device=dev), torch.empty_like(x)
    inv, cs, tp = torch.empty(B, device=dev), torch.empty(B, HW, device=dev), torch.empty(B, C // 64, device=dev)
    nws = lib.hk_bcnn_pool_ws_bytes(B, C, HW)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    assert lib.hk_bcnn_colsum_norm(p(x), p(cs), p(inv), B, C, HW, p(ws), nws, st()) == 0
    assert lib.hk_bcnn_gram_norm(p(x), p(inv), p(y), B, C, HW, st()) == 0
    bw = lambda: lib.hk_bcnn_bwd_gemm(p(x), p(y), p(dy), p(inv), p(dx), p(tp), B, C, HW, st())
    items = [(f'bwd_gemm bwd_v={v}', dict(bwd_v=v), bw) for v in (9, 0)]
    fl = 2.0 * B * C * C * HW
    out = [run_group('BCNN backward GEMM B=64 C=512 14x14', items, flops=fl)]
    items = [('colsum_norm', {}, lambda: lib.hk_bcnn_colsum_norm(p(x), p(cs), p(inv), B, C, HW, p(ws), nws, st()))]
    out.append(run_group('BCNN colsum+norm', items, bytes_=4.0 * B * C * HW))
    items = [('gram_norm', {}, lambda: lib.hk_bcnn_gram_norm(p(x), p(inv), p(y), B, C, HW, st()))]
    out.append(run_group('BCNN Gram fwd', items, flops=fl))
    y2, inv2, cs2 = torch.empty_like(y), torch.empty_like(inv), torch.empty_like(cs)

    def stages():
        lib.hk_bcnn_colsum_norm(p(x), p(cs2), p(inv2), B, C, HW, p(ws), nws, st())
        return lib.hk_bcnn_gram_norm(p(x), p(inv2), p(y2), B, C, HW, st())
    items = [('hk_bcnn_pool_fwd (column-sum partials + Gram with the norm in its prologue)', {},
              lambda: lib.hk_bcnn_pool_fwd(p(x), p(y2), p(inv2), p(cs2), B, C, HW, p(ws), nws, st())),
             ('hk_bcnn_colsum_norm + hk_bcnn_gram_norm (three launches)', {}, stages)]
    out.append(run_group('BCNN pool forward, whole', items, flops=fl))
    items = [('rank1', {}, lambda: lib.hk_bcnn_bwd_rank1(p(dx), p(tp), p(inv), p(cs), B, C, HW, st()))]
    out.append(run_group('BCNN rank-1 fix', items, bytes_=8.0 * B * C * HW))
    return out


def g_ssqrt():
    B, C, HW = 64, 512, 196
    x = torch.randn(B, C, HW, device=dev)
    y, dy, dx = torch.empty(B, C * C, device=dev), torch.randn(B, C * C, device=dev), torch.empty_like(x)
    inv = torch.empty(B, device=dev)
    nws = lib.hk_bcnn_ssqrt_ws_bytes(B, C, HW)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    fw = lambda: lib.hk_bcnn_ssqrt_pool_fwd(p(x), p(y), p(inv), B, C, HW, p(ws), nws, st())
    assert fw() == 0
    bw = lambda: lib.hk_bcnn_ssqrt_pool_bwd(p(x), p(y), p(dy), p(inv), p(dx), B, C, HW, p(ws), nws, st())
    fl = 2.0 * B * C * C * HW
    return [run_group('signed-sqrt pool B=64', [('fwd', {}, fw)] + [(f'bwd bwd_v={v}', dict(bwd_v=v), bw) for v in (9, 11)], flops=fl)]


def g_cov():
    B, C, HW = 64, 256, 196
    x = torch.relu(torch.randn(B, C, HW, device=dev))
    cov, mu, g, dx = torch.empty(B, C, C, device=dev), torch.empty(B, C, device=dev), torch.randn(B, C, C, device=dev), torch.empty_like(x)
    fw = lambda: lib.hk_cov_pool_fwd(p(x), p(cov), p(mu), B, C, HW, st())
    assert fw() == 0
    bw = lambda: lib.hk_cov_pool_bwd(p(x), p(mu), p(g), p(dx), B, C, HW, st())
    fl = 2.0 * B * C * C * HW
    items = [('fwd (centring in LDS)', {}, fw)] + [(f'bwd bwd_v={v}', dict(bwd_v=v), bw) for v in (0, 11, 12, 13, 14, 21, 1)]
    return [run_group('covariance B=64 C=256 14x14', items, flops=fl)]


def g_cbp():
    out = []
    C, HW, D = 512, 196, 6000
    plan = F.CbpPlan(*F.sketch_hashes(C, C, D), D, dev)
    for B in (64, 16):
        x = torch.relu(torch.randn(B, C, HW, device=dev))
        y, cr, inv = torch.empty(B, D, device=dev), torch.empty(B, D, device=dev), torch.empty(B, device=dev)
        dy, dx = torch.randn(B, D, device=dev), torch.empty_like(x)
        nws = lib.hk_cbp_ws_bytes(B, C, HW, D)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        fw = lambda: lib.hk_cbp_fwd(p(x), p(plan.blob), p(y), p(cr), p(inv), B, C, HW, D, p(ws), nws, st())
        assert fw() == 0
        bw = lambda: lib.hk_cbp_bwd(p(x), p(plan.blob), p(y), p(cr), p(inv), p(dy), p(dx), B, C, HW, D, p(ws), nws, st())
        items = [(f'fwd cbp_bin={v}', dict(cbp_bin=v), fw) for v in CBP_FWD] + [(f'bwd bwd_v={v}', dict(bwd_v=v), bw) for v in CBP_BWD]
        out.append(run_group(f'CBP B={B} C=512 14x14 D=6000', items, flops=2.0 * B * C * C * HW))
    return out


CBP_FWD = (3,)
CBP_BWD = (0,)


def g_ns():
    B, d, it = 64, 256, 5
    x = torch.relu(torch.randn(B, d, 14, 14, device=dev))
    a = F.covpool(x).detach().contiguous()
    out_, na = torch.empty_like(a), torch.empty(B, device=dev)
    ys, zs = torch.empty(B, it - 1, d, d, device=dev), torch.empty(B, it - 1, d, d, device=dev)
    g, da = torch.randn(B, d, d, device=dev), torch.empty_like(a)
    nf, nb = lib.hk_ns_sqrtm_ws_bytes(B, d, it, 0), lib.hk_ns_sqrtm_ws_bytes(B, d, it, 1)
    wf, wb = torch.empty(nf, dtype=torch.uint8, device=dev), torch.empty(nb, dtype=torch.uint8, device=dev)
    fw = lambda: lib.hk_ns_sqrtm_fwd(p(a), p(out_), p(na), p(ys), p(zs), B, d, it, p(wf), nf, st())
    assert fw() == 0
    bw = lambda: lib.hk_ns_sqrtm_bwd(p(a), p(out_), p(na), p(ys), p(zs), p(g), p(da), B, d, it, p(wb), nb, st())
    res = []
    fs = lambda: lib.hk_ns_sqrtm_fwd_sym(p(a), p(out_), p(na), p(ys), p(zs), B, d, it, p(wf), nf, st())
    items = [(f'fwd ns_streams={v}', dict(ns_streams=v), fw) for v in NS_STREAMS]
    items += [(f'fwd_sym ns_streams={v} ns_tn={tn}', dict(ns_streams=v, ns_tn=tn), fs) for v in (0, 1) for tn in (64,)]
    items += [(f'fwd_sym ns_flow={v} (one dataflow launch)', dict(ns_flow=v), fs) for v in (1, 2)]
    res.append(run_group('Newton-Schulz fwd B=64 d=256 it=5', items, flops=12 * 2.0 * d ** 3 * B))
    res.append(run_group('Newton-Schulz bwd B=64 d=256 it=5', [(f'bwd ns_streams={v}', dict(ns_streams=v), bw) for v in NS_STREAMS],
                         flops=38 * 2.0 * d ** 3 * B))
    tv, dtv = torch.empty(B, d * (d + 1) // 2, device=dev), torch.randn(B, d * (d + 1) // 2, device=dev)
    res.append(run_group('triuvec', [('fwd', {}, lambda: lib.hk_triu_vec_fwd(p(a), p(tv), B, d, st()))], bytes_=4.0 * B * d * (d + 1)))
    res.append(run_group('triuvec', [('bwd', {}, lambda: lib.hk_triu_vec_bwd(p(dtv), p(da), B, d, st()))],
                         bytes_=4.0 * B * (d * (d + 1) // 2 + d * d)))
    return res


NS_STREAMS = (1, 0)


def g_linear():
    res = []
    for tag, B, J, K in (('bcnn 262144->200', 64, 262144, 200), ('osme 100352->1024', 10, 100352, 1024), ('mpn 32896->200', 64, 32896, 200)):
        y, w, bias = torch.randn(B, J, device=dev), torch.randn(K, J, device=dev) * 0.01, torch.randn(K, device=dev)
        o, g = torch.empty(B, K, device=dev), torch.randn(B, K, device=dev)
        dyy, dw, db = torch.empty_like(y), torch.empty_like(w), torch.empty(K, device=dev)
        nws = lib.hk_linear_ws_bytes(B, J, K)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        fw = lambda: lib.hk_linear_fwd(p(y), p(w), p(bias), p(o), B, J, K, p(ws), nws, st())
        bw = lambda: lib.hk_linear_bwd(p(y), p(w), p(g), p(dyy), p(dw), p(db), B, J, K, st())

        def tfw():
            torch.nn.functional.linear(y, w, bias)

        def tbw():
            g @ w
            g.t() @ y
            g.sum(0)
        by = 4.0 * (K * J + B * J)
        items = [('hk_linear_fwd', {}, fw)]
        res.append(run_group(f'linear fwd {tag}', items + [('torch (rocBLAS)', {}, tfw)], flops=2.0 * B * J * K, bytes_=by))
        res.append(run_group(f'linear bwd {tag}', [('hk_linear_bwd', {}, bw), ('torch (rocBLAS x3)', {}, tbw)], flops=4.0 * B * J * K, bytes_=2 * by))
        del y, w, dyy, dw
    return res


def g_cin():
    B, C, HW = 20, 2048, 49
    x = torch.relu(torch.randn(B, C, HW, device=dev))
    w, y = torch.empty(B, C, C, device=dev), torch.empty(B, C, HW, device=dev)
    fw = lambda: lib.hk_cin_sci_fwd(p(x), p(w), p(y), B, C, HW, st())
    assert fw() == 0
    wr = torch.softmax(torch.bmm(x, x.transpose(1, 2)).mul_(-1.0 / HW), dim=2)
    yr = torch.bmm(wr, x)
    err = (float((w - wr).norm() / wr.norm()), float((y - yr).norm() / yr.norm()))

    def tfw():
        w_ = torch.softmax(torch.bmm(x, x.transpose(1, 2)).mul_(-1.0 / HW), dim=2)
        torch.bmm(w_, x)
    r = run_group('CIN SCI forward B=20 C=2048 7x7', [('hk_cin_sci_fwd (one kernel)', {}, fw), ('hk_cin_sci_fwd three-kernel chain', dict(bcnn_generic=1), fw),
                                                    ('torch bmm + softmax + bmm', {}, tfw)], flops=2 * 2.0 * B * C * C * HW, bytes_=4.0 * B * C * C)
    r['rel_err_w_y_vs_torch'] = err
    return [r]


GROUPS = {'cin': g_cin, 'bcnn': g_bcnn, 'ssqrt': g_ssqrt, 'cov': g_cov, 'cbp': g_cbp, 'ns': g_ns, 'linear': g_linear}

if __name__ == '__main__':
    which = [a for a in sys.argv[1:] if a in GROUPS] or list(GROUPS)
    out = {'device': torch.cuda.get_device_name(0), 'rounds': ROUNDS, 'iters': ITERS, 'groups': []}
    for w in which:
        try:
            out['groups'] += GROUPS[w]()
        except Exception as e:            # a failing group must not lose the others' numbers
            out['groups'].append({'group': w, 'error': repr(e)})
        torch.cuda.synchronize()
    s = json.dumps(out, indent=1)
    print(s)
    od = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(od):
        open(os.path.join(od, 'r3_lab.json'), 'w').write(s)
