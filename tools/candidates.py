"""HIP-event timings of the opt-in variants and SURVEY-8f rows that were written after round 1's GPU budget was spent,
each next to the path it would replace.  bench.py runs this in a subprocess (after the headline measurement, with a
timeout; its failure cannot touch the headline) and attaches the rows as `candidates`, so the driver's round-end
bench run doubles as their first measurement.
    python tools/candidates.py [--step]        # prints one JSON list
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as TF

from hawkeye_amd.miopen_cache import use_in_tree_cache

use_in_tree_cache()
import hawkeye_amd.functional as F
from hawkeye_amd import _lib
from hawkeye_amd._lib import ptr, stream

# HK_CAND_TINY=1 without a GPU: dry run of this script on the CPU emulation of the kernels (tests/emu) with shrunken
# shapes - checks every call's argument marshalling before the script runs unattended on the GPU box.
TINY = os.environ.get('HK_CAND_TINY') == '1'
if TINY and not torch.cuda.is_available():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
    from emu import harness
    _ctx = harness.emulated()
    _ctx.__enter__()                                    # stays active for the life of the process
    lib, ptr, stream, dev = harness.load_emu(), harness._cpu_ptr, (lambda: None), torch.device('cpu')
else:
    lib = _lib.load()
    dev = torch.device('cuda:0')
rows = []


def knob(name, value):
    """A/B lever of the library (hk_tuning_set); every function below restores what it touches."""
    rc = lib.hk_tuning_set(name.encode(), int(value))
    if rc != 0:
        raise RuntimeError(f'hk_tuning_set({name}) returned {rc}')


def sz(full, tiny):
    return tiny if TINY else full


_first_round_us = [None]


def timeit(fn, iters=20, warm=3, rounds=3):
    """HIP-event time per call (us) of the LAST of `rounds` back-to-back rounds of `iters` calls - the chip's clock settles
    only after ~10 ms of load, so the first round of a variant reads up to 15 % slower than the same code a moment later
    (BENCH_r02: 313 vs 268 us for two rows that were the same dispatch).  The first round's value is kept for the row."""
    if TINY:
        iters, warm, rounds = 1, 1, 1
    for _ in range(warm):
        rc = fn()
        if isinstance(rc, int) and rc != 0:
            raise RuntimeError(f'C ABI call returned {rc}')
    if dev.type != 'cuda':
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        _first_round_us[0] = None
        return (time.perf_counter() - t0) / iters * 1e6
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(rounds + 1)]
    torch.cuda.synchronize()
    ev[0].record()
    for r in range(rounds):
        for _ in range(iters):
            fn()
        ev[r + 1].record()
    torch.cuda.synchronize()
    _first_round_us[0] = ev[0].elapsed_time(ev[1]) / iters * 1e3
    return ev[rounds - 1].elapsed_time(ev[rounds]) / iters * 1e3


def row(op, variant, us, flops=0.0, bytes_=0.0, note=''):
    r = {'op': op, 'variant': variant, 'us': round(us, 1), 'tflops': round(flops / us / 1e6, 1),
         'gbs': round(bytes_ / us / 1e3), 'note': note}
    if _first_round_us[0] is not None:
        r['us_first_round'] = round(_first_round_us[0], 1)
        _first_round_us[0] = None
    rows.append(r)


def guarded(fn):
    try:
        fn()
    except Exception as e:  # noqa: BLE001
        rows.append({'op': fn.__name__, 'error': repr(e)[:300]})


def linear():
    shapes = {'bcnn 262144->200': (64, 262144, 200), 'mpn 32896->200': (64, 32896, 200),
              'osme 100352->1024 N=10': (10, 100352, 1024)}
    if TINY:
        shapes = {'bcnn 262144->200': (5, 640, 70), 'osme 100352->1024 N=10': (3, 200, 9)}
    for tag, (B, J, K) in shapes.items():
        y = torch.randn(B, J, device=dev)
        w = torch.randn(K, J, device=dev) * 0.01
        b = torch.zeros(K, device=dev)
        g = torch.randn(B, K, device=dev)
        out = torch.empty(B, K, device=dev)
        dy, dw, db = torch.empty_like(y), torch.empty_like(w), torch.empty_like(b)
        nws = lib.hk_linear_ws_bytes(B, J, K)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        fl = 2.0 * B * J * K
        row(f'linear fwd {tag}', 'hk_linear_fwd (automatic: wide-classifier kernel where it applies)',
            timeit(lambda: lib.hk_linear_fwd(ptr(y), ptr(w), ptr(b), ptr(out), B, J, K, ptr(ws), nws, stream())), fl,
            4.0 * (B * J + K * J))
        row(f'linear fwd {tag}', 'torch F.linear (rocBLAS/hipBLASLt)', timeit(lambda: TF.linear(y, w, b)), fl,
            4.0 * (B * J + K * J))
        row(f'linear bwd {tag}', 'hk_linear_bwd (dy + dW + db)',
            timeit(lambda: lib.hk_linear_bwd(ptr(y), ptr(w), ptr(g), ptr(dy), ptr(dw), ptr(db), B, J, K, stream())), 2 * fl,
            4.0 * (2 * B * J + 2 * K * J))
        row(f'linear bwd {tag}', 'torch (g @ w, g.t() @ y, g.sum(0))', timeit(lambda: (g @ w, g.t() @ y, g.sum(0))), 2 * fl,
            4.0 * (2 * B * J + 2 * K * J))
        err = float((out - TF.linear(y, w, b)).norm() / TF.linear(y, w, b).norm())
        rows[-4]['rel_err_vs_torch'] = err
        if tag.startswith('bcnn'):                      # -1: the generic split-K path (round-2a default, 256 slabs); then a
            for slabs in sz((-1, 128, 192, 256, 512), (2, 3)):      # slab-count sweep of the wide-classifier kernel (auto = 256)
                knob('linear_slabs', slabs)
                n2 = lib.hk_linear_ws_bytes(B, J, K)
                ws2 = torch.empty(n2, dtype=torch.uint8, device=dev)
                row(f'linear fwd {tag}', f'hk_linear_fwd linear_slabs={slabs}',
                    timeit(lambda: lib.hk_linear_fwd(ptr(y), ptr(w), ptr(b), ptr(out), B, J, K, ptr(ws2), n2, stream())), fl,
                    4.0 * (B * J + K * J))
            knob('linear_slabs', 0)


def ns_chain():
    B, d = sz(64, 2), sz(256, 128)
    x = torch.relu(torch.randn(B, d, 196, device=dev))
    cov, mu = torch.empty(B, d, d, device=dev), torch.empty(B, d, device=dev)
    lib.hk_cov_pool_fwd(ptr(x), ptr(cov), ptr(mu), B, d, 196, stream())
    out, na = torch.empty(B, d, d, device=dev), torch.empty(B, device=dev)
    ys, zs = torch.empty(B, 4, d, d, device=dev), torch.empty(B, 4, d, d, device=dev)
    g, da = torch.randn(B, d, d, device=dev).triu(), torch.empty(B, d, d, device=dev)
    nwf, nwb = lib.hk_ns_sqrtm_ws_bytes(B, d, 5, 0), lib.hk_ns_sqrtm_ws_bytes(B, d, 5, 1)
    wf, wb = torch.empty(nwf, dtype=torch.uint8, device=dev), torch.empty(nwb, dtype=torch.uint8, device=dev)
    ref = None
    for tn in (0, 64, 128):
        knob('ns_tn', tn)
        f = timeit(lambda: lib.hk_ns_sqrtm_fwd(ptr(cov), ptr(out), ptr(na), ptr(ys), ptr(zs), B, d, 5, ptr(wf), nwf, stream()))
        b = timeit(lambda: lib.hk_ns_sqrtm_bwd(ptr(cov), ptr(out), ptr(na), ptr(ys), ptr(zs), ptr(g), ptr(da), B, d, 5,
                                               ptr(wb), nwb, stream()))
        tag = 'grouped products, ' + ('automatic tile width (default)' if tn == 0 else f'ns_tn={tn}')
        row('ns_sqrtm fwd B=64 d=256 it=5', tag, f, 12 * 2.0 * B * d ** 3)
        row('ns_sqrtm bwd B=64 d=256 it=5', tag, b, 38 * 2.0 * B * d ** 3)
        if ref is None:
            ref = (out.clone(), da.clone())
        else:
            rows[-2]['rel_vs_default'] = float((out - ref[0]).norm() / ref[0].norm())
            rows[-1]['rel_vs_default'] = float((da - ref[1]).norm() / ref[1].norm())
    knob('ns_tn', 0)
    # what the MPN head calls (its input is a covariance): tiles below the diagonal blocks mirrored, not computed
    f = timeit(lambda: lib.hk_ns_sqrtm_fwd_sym(ptr(cov), ptr(out), ptr(na), ptr(ys), ptr(zs), B, d, 5, ptr(wf), nwf, stream()))
    row('ns_sqrtm fwd B=64 d=256 it=5', 'hk_ns_sqrtm_fwd_sym (symmetric input: 3 of 4 tiles per product; the MPN head)', f,
        12 * 2.0 * B * d ** 3)
    rows[-1]['rel_vs_default'] = float((out - ref[0]).norm() / ref[0].norm())


def npairs():
    for b, p, D in sz(((10, 2, 1024), (32, 2, 1024)), ((4, 2, 24),)):
        x = torch.randn(b, p, D, device=dev)
        t = torch.arange(b, device=dev) // 2
        labels = t.to(torch.int32)
        loss, dx = torch.empty(1, device=dev), torch.empty_like(x)
        nws = lib.hk_npairs_ws_bytes(b * p, D)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        row(f'npairs loss+grad b={b} p={p} D={D}', 'hk_npairs_loss (5 launches)',
            timeit(lambda: lib.hk_npairs_loss(ptr(x), ptr(labels), ptr(loss), ptr(dx), b, p, D, ptr(ws), nws, stream())))

        def torch_vectorised():
            xr = x.clone().requires_grad_(True)
            n = b * p
            xn = TF.normalize(xr.view(n, -1))
            s = xn @ xn.t()
            cls, att = torch.repeat_interleave(t, p), torch.arange(p, device=dev).repeat(b)
            sc, sa = cls[:, None] == cls[None, :], att[:, None] == att[None, :]
            diff = s[:, None, :] - s[:, :, None]

            def term(pos, neg):
                return (torch.log(1 + (torch.exp(diff) * neg[:, None, :]).sum(2)) * pos).sum()
            l = (term(sc & sa, ~(sc & sa)) + term(~sc & sa, ~sc & ~sa) + term(sc & ~sa, ~sc & ~sa)) / n
            l.backward()
            return l
        row(f'npairs loss+grad b={b} p={p} D={D}', 'torch vectorised fwd+bwd (not the reference python loop)',
            timeit(torch_vectorised, iters=10))
        rows[-2]['loss'] = float(loss)
        rows[-1]['loss'] = float(torch_vectorised())


def cbp():
    C, HW, D = sz(512, 128), 196, sz(6000, 1024)
    plan = F.CbpPlan(*F.sketch_hashes(C, C, D), D, dev)
    for B in sz((64, 16), (2,)):                        # the metric's batch and the yaml's (configs/CBCNN_S2.yaml:10)
        x = torch.relu(torch.randn(B, C, HW, device=dev))
        y, craw, inv = torch.empty(B, D, device=dev), torch.empty(B, D, device=dev), torch.empty(B, device=dev)
        nws = lib.hk_cbp_ws_bytes(B, C, HW, D)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        ref = None
        for flag, tag in (('0', 'Gram + row-sketch binning (round-1 default at B=64)'), ('1', 'Gram + CSR gather binning (round-1 default at B=16)'),
                          ('2', 'Gram + row-scatter binning (round-2 default)'),
                          ('3', 'Gram and binning fused, G never written (hk_cbp_fused.h; round-3 default)')):
            knob('cbp_bin', flag)
            row(f'cbp fwd B={B}', tag,
                timeit(lambda: lib.hk_cbp_fwd(ptr(x), ptr(plan.blob), ptr(y), ptr(craw), ptr(inv), B, C, HW, D, ptr(ws), nws,
                                              stream())), 2.0 * B * C * C * HW)
            if flag == '0':
                ref = y.clone()
            elif flag == '2':
                rows[-1]['bit_identical_to_row_sketch'] = bool(torch.equal(y, ref))
            elif flag == '3':
                rows[-1]['rel_vs_row_sketch'] = float((y - ref).norm() / ref.norm())
        knob('cbp_bin', -1)
        dy, dx = torch.randn(B, D, device=dev), torch.empty_like(x)
        ref = None
        for flag, tag in ((1, 'bwd_v=1: dc kernel + 64-row panel kernel (round 1)'), (4, 'bwd_v=4: dc kernel + eight-wave 64-row kernel'),
                          (5, 'bwd_v=5: dc kernel + register-staged 128-row kernel (round-2 default at B=64)'),
                          (0, 'hk_bwd3c.h: dc formed in the kernel, P generated in LDS, X by LDS-DMA (round-3 default)')):
            knob('bwd_v', flag)
            row(f'cbp bwd B={B}', tag,
                timeit(lambda: lib.hk_cbp_bwd(ptr(x), ptr(plan.blob), ptr(y), ptr(craw), ptr(inv), ptr(dy), ptr(dx), B, C, HW, D,
                                              ptr(ws), nws, stream())), 2.0 * B * C * C * HW)
            if ref is None:
                ref = dx.clone()
            else:
                rows[-1]['rel_vs_64row'] = float((dx - ref).norm() / ref.norm())
        knob('bwd_v', 0)
    knob('cbp_bin', -1)


def bwd_variants():
    B, C, HW = sz(64, 2), sz(512, 128), 196
    x = torch.relu(torch.randn(B, C, HW, device=dev))
    y, dy, dx = torch.empty(B, C * C, device=dev), torch.randn(B, C * C, device=dev), torch.empty_like(x)
    inv, cs, tp = torch.empty(B, device=dev), torch.empty(B, HW, device=dev), torch.empty(B, C // 64, device=dev)
    nws = lib.hk_bcnn_pool_ws_bytes(B, C, HW)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    lib.hk_bcnn_colsum_norm(ptr(x), ptr(cs), ptr(inv), B, C, HW, ptr(ws), nws, stream())
    lib.hk_bcnn_gram_norm(ptr(x), ptr(inv), ptr(y), B, C, HW, stream())
    dc = sz(256, 64)
    xm = torch.relu(torch.randn(B, dc, HW, device=dev))
    mu, g, dxm = torch.zeros(B, dc, device=dev), torch.randn(B, dc, dc, device=dev), torch.empty_like(xm)
    ref = None
    for flag in (1, 4, 5, 9, 14, 12, 13, 11, 15, 0):
        knob('bwd_v', flag)
        tag = {1: 'bwd_v=1: 64-row blocks, P tile built in LDS, 2 WGs/CU (round-1 kernel)',
               4: 'bwd_v=4: 64-row blocks on the eight-wave raw-tile kernel (round-2 default for the covariance at C=256, B=64)',
               5: 'bwd_v=5: 128-row blocks, raw tiles, one barrier per K-block, 1 WG/CU (hk_bwd128.h)',
               9: 'bwd_v=9: 128-row blocks staged by LDS-DMA, swizzled tiles (hk_bwd128d.h; round-2 default, BCNN mode only)',
               14: 'bwd_v=14: hk_bwd3.h without its two changes (= bwd_v 9 for BCNN; LDS-DMA + mu column for the covariance)',
               12: 'bwd_v=12: hk_bwd3.h, remainder columns on the VALU only',
               13: 'bwd_v=13: hk_bwd3.h, LDS-staged epilogue only',
               11: 'bwd_v=11: hk_bwd3.h, VALU remainder columns + LDS-staged epilogue',
               15: 'bwd_v=15: as 11 + a wave per 16 rows and all column tiles (128-row blocks only: same as 11 for the covariance)',
               0: 'hk_bwd3.h, automatic (round-3 default): as 15 + late coefficient at 128-row blocks (BCNN); bwd_v 11 form at 64-row blocks (covariance)'}[flag]
        row('bcnn bwd_gemm B=64 C=512', tag,
            timeit(lambda: lib.hk_bcnn_bwd_gemm(ptr(x), ptr(y), ptr(dy), ptr(inv), ptr(dx), ptr(tp), B, C, HW, stream()), iters=40),
            2.0 * B * C * C * HW, 8.0 * B * (C * C + C * HW))
        if ref is None:
            ref = dx.clone()
        else:
            rows[-1]['rel_vs_64row'] = float((dx - ref).norm() / ref.norm())
        row('cov_pool bwd B=64 C=256', tag,
            timeit(lambda: lib.hk_cov_pool_bwd(ptr(xm), ptr(mu), ptr(g), ptr(dxm), B, dc, HW, stream()), iters=40),
            2.0 * B * dc * dc * HW)
    knob('bwd_v', 0)


def roi_bwd():
    B, C = sz(16, 2), sz(512, 8)
    dy, dx = torch.randn(B, C, 56, 56, device=dev), torch.empty(B, C, 56, 56, device=dev)
    drop = torch.tensor([[10., 12., 20., 30.]] * B, device=dev)
    # a 37 x 28 crop (tap windows up to 3 x 3), a 23 x 21 one (4 x 4) and a 9 x 11 one (table loops)
    for bx, tag in (([3.2, 5.9, 40.1, 33.3], '37x28 crop'), ([20.2, 14.9, 43.1, 35.3], '23x21 crop'), ([20.2, 14.9, 29.1, 25.3], '9x11 crop')):
        box = torch.tensor([bx] * B, device=dev)
        ref = None
        for flag in (1, 0):
            knob('roi_bwd', flag)
            row(f'roi_crop_resize bwd B={B} C={C} 56x56, {tag}', 'roi_bwd=1: round-1 table kernel' if flag == 1 else
                'crop-only gather, compile-time 3x3 / 4x4 windows with register weights, LDS output image (default)',
                timeit(lambda: lib.hk_roi_crop_resize_bwd(ptr(dy), ptr(box), ptr(drop), ptr(dx), B, C, 56, 56, 1, stream())), 0.0,
                8.0 * B * C * 3136)
            if ref is None:
                ref = dx.clone()
            else:
                rows[-1]['bit_identical_to_default'] = bool(torch.equal(dx, ref))
        ref = None
        for flag in (1, 0):                                 # (the knob keeps the round-1 kernels of both directions)
            knob('roi_bwd', flag)
            row(f'roi_crop_resize fwd B={B} C={C} 56x56, {tag}', 'roi_bwd=1: round-1 kernel, one map per workgroup, taps from global memory' if flag == 1 else
                'LDS-staged maps, 16 maps per workgroup, per-pixel geometry in registers (default)',
                timeit(lambda: lib.hk_roi_crop_resize_fwd(ptr(dy), ptr(box), ptr(drop), ptr(dx), B, C, 56, 56, 1, stream())), 0.0,
                8.0 * B * C * 3136)
            if ref is None:
                ref = dx.clone()
            else:
                rows[-1]['bit_identical_to_default'] = bool(torch.equal(dx, ref))
    knob('roi_bwd', 0)


def cin():
    B, C, HW = sz(20, 2), sz(2048, 96), 49                                   # configs/CIN.yaml: 4 classes x 5 samples, ResNet-50 7x7 map
    x = torch.relu(torch.randn(B, C, HW, device=dev))
    wt = torch.randn(B, device=dev)
    w, y, yc = torch.empty(B, C, C, device=dev), torch.empty_like(x), torch.empty_like(x)
    dy, dx, dx2 = torch.randn_like(x), torch.empty_like(x), torch.empty_like(x)
    dw, dwbuf, dwt = torch.empty_like(w), torch.empty_like(w), torch.empty(B, device=dev)
    nws = lib.hk_cin_cci_ws_bytes(B, C)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    fl = 2.0 * B * C * C * HW
    row('cin sci fwd B=20 C=2048 HW=49', 'hk_cin_sci_fwd (one flash-style kernel: statistics pass + normalised second product, W written once; round-3 default)',
        timeit(lambda: lib.hk_cin_sci_fwd(ptr(x), ptr(w), ptr(y), B, C, HW, stream())), 2 * fl, 4.0 * B * (2 * C * C))

    def torch_sci():
        ws_ = torch.softmax(-torch.bmm(x, x.transpose(1, 2)) / HW, dim=2)
        return torch.bmm(ws_, x)
    knob('bcnn_generic', 1)
    row('cin sci fwd B=20 C=2048 HW=49', 'bcnn_generic=1: Gram kernel + row softmax + second product (round-2 chain)',
        timeit(lambda: lib.hk_cin_sci_fwd(ptr(x), ptr(w), ptr(y), B, C, HW, stream())), 2 * fl, 4.0 * B * (4 * C * C))
    knob('bcnn_generic', 0)
    lib.hk_cin_sci_fwd(ptr(x), ptr(w), ptr(y), B, C, HW, stream())
    row('cin sci fwd B=20 C=2048 HW=49', 'torch bmm + softmax + bmm (reference)', timeit(torch_sci), 2 * fl,
        4.0 * B * (4 * C * C))
    rows[-3]['rel_err_vs_torch'] = float((y - torch_sci()).norm() / torch_sci().norm())
    row('cin cci fwd', 'hk_cin_cci_fwd (|W - w W\'| in the operand loader)',
        timeit(lambda: lib.hk_cin_cci_fwd(ptr(x), ptr(w), ptr(wt), ptr(yc), B, C, HW, stream())), fl)
    row('cin cci bwd', 'hk_cin_cci_bwd',
        timeit(lambda: lib.hk_cin_cci_bwd(ptr(x), ptr(w), ptr(wt), ptr(dy), ptr(dx2), ptr(dw), ptr(dwt), B, C, HW, ptr(ws),
                                          nws, stream())), 2 * fl)
    row('cin sci bwd', 'hk_cin_sci_bwd (with the CCI gradient into W)',
        timeit(lambda: (dwbuf.copy_(dw), lib.hk_cin_sci_bwd(ptr(x), ptr(w), ptr(dy), ptr(dwbuf), 1, ptr(dx), B, C, HW,
                                                             stream()))[1]), 3 * fl)


def bcnn_step_with_hip_linear():
    import hawkeye_amd.model  # noqa: F401
    from hawkeye_amd.config import CfgNode
    from hawkeye_amd.model.registry import MODEL
    torch.manual_seed(0)
    m = MODEL.get('BCNN')(CfgNode(dict(name='BCNN', stage=2, num_classes=200))).to(dev).to(memory_format=torch.channels_last)
    m.train()
    opt = torch.optim.SGD(m.parameters(), lr=0.005, momentum=0.9, weight_decay=1e-5)
    crit = torch.nn.CrossEntropyLoss(label_smoothing=0.1)
    x = torch.randn(64, 3, 448, 448, device=dev).contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, 200, (64,), device=dev)

    def step():
        loss = crit(m(x), y)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    for flag in ('0', '1'):
        os.environ['HAWKEYE_HIP_LINEAR'] = flag
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 8 * 1e3
        rows.append({'op': 'BCNN bs64 448^2 train step', 'variant': f'HAWKEYE_HIP_LINEAR={flag}', 'ms_per_step': round(ms, 2),
                     'images_per_sec': round(64 / ms * 1e3, 1)})
    os.environ.pop('HAWKEYE_HIP_LINEAR', None)


if __name__ == '__main__':
    for f in (bwd_variants, roi_bwd, linear, ns_chain, npairs, cbp, cin):
        guarded(f)
    if '--step' in sys.argv and not TINY:
        guarded(bcnn_step_with_hip_linear)
    print(json.dumps(rows), flush=True)
    if TINY and dev.type != 'cuda':
        _ctx.__exit__(None, None, None)
