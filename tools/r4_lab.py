"""Round-4 A/B harness: HIP-event timings of head kernels and the alternatives they are compared with, ALTERNATING over
several rounds in one process (clock state moves single measurements by several per cent; the first round of a process
is cold) - median, first-round and minimum per item.  Direct C-ABI calls on torch's current stream; torch / rocBLAS
calls where the row names them.
    python tools/r4_lab.py [group ...]        # groups: linear ; default: all
Prints one JSON object (also written to gpurun_out/r4_lab_<groups>.json when that directory exists)."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from hawkeye_amd import _lib

lib = _lib.load()
P = ctypes.c_void_p
dev = torch.device('cuda:0')
st = lambda: P(torch.cuda.current_stream().cuda_stream)
p = lambda t: P(t.data_ptr()) if t is not None else None
ROUNDS, ITERS = 5, 20
PEAK_TF, PEAK_GBS = 157.3, 8000.0
DEFAULTS = dict(bwd_v=0, cbp_bin=-1, ns_streams=1, ns_tn=0, linear_slabs=0, bcnn_generic=0, ns_sym=1, sched_b=0, lin_walk=-1)


def knobs(**kw):
    for k, v in kw.items():
        assert lib.hk_tuning_set(k.encode(), int(v)) == 0, k


def run_group(title, items, flops=None, bytes_=None):
    """items: list of (tag, knob dict, fn[, flops, bytes]) - fn() enqueues one call and returns its rc (or None)."""
    out = {it[0]: [] for it in items}
    for rnd in range(ROUNDS):
        for it in items:
            tag, kn, fn = it[:3]
            knobs(**DEFAULTS)
            knobs(**kn)
            for _ in range(3):
                rc = fn()
                assert rc in (0, None) or not isinstance(rc, int), (tag, rc)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(ITERS):
                fn()
            e1.record()
            torch.cuda.synchronize()
            out[tag].append(round(e0.elapsed_time(e1) / ITERS * 1e3, 2))
    knobs(**DEFAULTS)
    res = {}
    for it in items:
        tag = it[0]
        v = out[tag]
        fl = it[3] if len(it) > 3 else flops
        by = it[4] if len(it) > 4 else bytes_
        med = sorted(v)[len(v) // 2]
        r = {'us': med, 'us_first_round': v[0], 'us_min': min(v)}
        if fl:
            r['tflops'] = round(fl / med / 1e6, 1)
            r['frac_mfma'] = round(fl / med / 1e6 / PEAK_TF, 3)
        if by:
            r['gbs'] = round(by / med / 1e3, 0)
            r['frac_hbm'] = round(by / med / 1e3 / PEAK_GBS, 3)
        res[tag] = r
    return {'group': title, 'rows': res}


def g_linear():
    out = []
    shapes = [('BCNN 64 x 262144 -> 200', 64, 262144, 200), ('MPN 64 x 32896 -> 200', 64, 32896, 200),
              ('OSME 10 x 100352 -> 1024', 10, 100352, 1024), ('CBCNN-Gram 16 x 262144 -> 200', 16, 262144, 200)]
    if 'pitch' in sys.argv:       # is the 1 MB row pitch of BCNN's W (J = 2^18 floats) what holds the streams at 3.8 TB/s?
        shapes = [('BCNN 64 x 262144 -> 200', 64, 262144, 200), ('pitch + 4 KB: 64 x 263168 -> 200', 64, 263168, 200),
                  ('pitch + 16 KB: 64 x 266240 -> 200', 64, 266240, 200)]
    for name, B, J, K in shapes:
        y = torch.randn(B, J, device=dev)
        w = torch.randn(K, J, device=dev) * 0.01
        bias = torch.zeros(K, device=dev)
        g = torch.randn(B, K, device=dev)
        o = torch.empty(B, K, device=dev)
        dy, dw, db = torch.empty(B, J, device=dev), torch.empty(K, J, device=dev), torch.empty(K, device=dev)
        nws = lib.hk_linear_ws_bytes(B, J, K)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        fl, by = 2.0 * B * J * K, 4.0 * (K * J + B * J + B * K)
        fwd = lambda: lib.hk_linear_fwd(p(y), p(w), p(bias), p(o), B, J, K, p(ws), nws, st())
        bwd = lambda: lib.hk_linear_bwd(p(y), p(w), p(g), p(dy), p(dw), p(db), B, J, K, st())
        bwd_dy = lambda: lib.hk_linear_bwd(p(y), p(w), p(g), p(dy), None, None, B, J, K, st())
        bwd_dw = lambda: lib.hk_linear_bwd(p(y), p(w), p(g), None, p(dw), p(db), B, J, K, st())

        def lib_fwd():
            torch.addmm(bias, y, w.t(), out=o)

        def lib_bwd():
            torch.mm(g, w, out=dy)
            torch.mm(g.t(), y, out=dw)
            torch.sum(g, 0, out=db)
        items = [('hk_linear_fwd', {}, fwd, fl, by),
                 ('rocBLAS fwd (torch.addmm)', {}, lib_fwd, fl, by),
                 ('hk_linear_bwd (dy + dW + db)', {}, bwd, 2 * fl, 2 * by),
                 ('hk_linear_bwd, contiguous slabs (lin_walk=0)', dict(lin_walk=0), bwd, 2 * fl, 2 * by),
                 ('hk_linear_bwd, interleaved chunks (lin_walk=1)', dict(lin_walk=1), bwd, 2 * fl, 2 * by),
                 ('hk_linear_bwd generic tiles (linear_slabs=-1)', dict(linear_slabs=-1), bwd, 2 * fl, 2 * by),
                 ('hk_linear_bwd dy only', {}, bwd_dy, fl, by),
                 ('hk_linear_bwd dW + db only', {}, bwd_dw, fl, by),
                 ('rocBLAS bwd (torch.mm x 2 + sum)', {}, lib_bwd, 2 * fl, 2 * by)]
        out.append(run_group('classifier ' + name, items))
        del y, w, dy, dw
        torch.cuda.empty_cache()
    return out


GROUPS = {'linear': g_linear}

if __name__ == '__main__':
    want = [a for a in sys.argv[1:] if a in GROUPS] or list(GROUPS)
    res = {'device': torch.cuda.get_device_name(0), 'rounds': ROUNDS, 'iters': ITERS, 'groups': []}
    for gname in want:
        res['groups'] += GROUPS[gname]()
    txt = json.dumps(res, indent=1)
    print(txt)
    if os.path.isdir(os.path.join(ROOT, 'gpurun_out')):
        with open(os.path.join(ROOT, 'gpurun_out', 'r4_lab_' + '_'.join(want) + '.json'), 'w') as f:
            f.write(txt)
