"""The other BASELINE.json configs on ONE MI355X (configs[2..4] at their single-GPU shapes), next to the headline:
    MPN   (Fast MPN-COV, ResNet-50, 448x448, batch 64)            configs/MPN.yaml
    CBCNN (VGG-16, 448x448, D = 6000, batch 64 and the yaml's 16)  configs/CBCNN_S2.yaml
    APCNN (ResNet-50 + FPN, 448x448, batch 16, 8142 classes = iNat2018 shape: hidden_num 256, border 0.1-0.9,
           model/methods/APCNN.py:360-363,451-454)                 configs/APCNN.yaml
For each: ms per full train step (forward, loss, backward, SGD step; synthetic batch resident in HBM) and the
HIP-event time + roofline fraction of every hand-written kernel of its pooling head at that batch size.
bench.py runs this in a subprocess after the headline measurement and attaches the list as `other_models`.
    python tools/model_rows.py [--quick | --kernels-only]      # prints one JSON list
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench                      # build_model, time_events, peaks (also points MIOpen at the in-tree cache)
import hawkeye_amd.functional as F
from hawkeye_amd import _lib
from hawkeye_amd._lib import ptr, stream

QUICK = '--quick' in sys.argv
dev = torch.device('cuda:0')
lib = _lib.load()
rows = []


def E(*shape, dtype=torch.float32):
    return torch.empty(*shape, dtype=dtype, device=dev)


def R(*shape):
    return torch.randn(*shape, device=dev)


_PMC = {}


def pmc_traffic(model_csv, kernels):
    """HBM bytes per launch sequence from the committed in-step counters of that config (profiles/r6_step_<cfg>_pmc.csv: FETCH_SIZE /
    WRITE_SIZE in KiB, separate --pmc passes, FETCH_SIZE doubled per the guide) summed over the row's kernels - `kernels` = a list of
    (kernel-name prefix, launches of it in the sequence).  None when a kernel is not in the file (counters cannot be sampled from
    inside this process: this is the last profiled value, like bench.pmc_traffic)."""
    import csv
    path = os.path.join(ROOT, 'profiles', model_csv)
    if path not in _PMC:
        try:
            _PMC[path] = list(csv.DictReader(open(path)))
        except OSError:
            _PMC[path] = []
    total = 0.0
    for prefix, count in kernels:
        hit = [r for r in _PMC[path] if r['Kernel'].replace('void ', '').startswith('hk::' + prefix)]
        by_kernel = {}
        for r in hit:
            if r['Counter'] in ('FETCH_SIZE', 'WRITE_SIZE'):
                by_kernel.setdefault(r['Kernel'], {})[r['Counter']] = float(r['MeanValue'])
        vals = [v for v in by_kernel.values() if len(v) == 2]
        if not vals:
            return None
        v = vals[0] if len(vals) == 1 else {k: sum(x[k] for x in vals) / len(vals) for k in ('FETCH_SIZE', 'WRITE_SIZE')}
        total += count * (2.0 * v['FETCH_SIZE'] + v['WRITE_SIZE']) * 1024.0
    return round(total)


def kernel_row(model, name, fn, flops=0.0, bytes_=0.0, iters=30, flops_exec=None, pmc=None):
    """flops / bytes_: algorithmic work per call; flops_exec: what the kernels issue when that is less (symmetric
    Newton-Schulz forward: 6 of 8 tiles per product; backward: 34 of the reference's 38 products) - bench.roof_row prices
    `frac` on the smaller of the two."""
    us = bench.time_events(fn, iters, rounds=3)[0] * 1e3           # third back-to-back round: settled clocks (bench.py)
    r = bench.roof_row(name, us, flops, bytes_, flops_exec)
    r['model'] = model
    if pmc is not None:                    # (csv of the config's in-step counters, [(kernel prefix, launches)])
        r['traffic'] = pmc_traffic(*pmc)
        if r['traffic'] is not None:
            r['traffic_source'] = 'profiles/' + pmc[0]
            if bytes_:
                r['traffic_over_algorithmic'] = round(r['traffic'] / bytes_, 3)
    rows.append(r)
    if us < 12.0:     # back-to-back calls through python + ctypes cost ~8-9 us each: below that the row times the host
        r['note'] = 'host-call-bound row (python + ctypes ~ 9 us per call): rocprofv3 durations in profiles/r4_step_*_kernel_stats.csv'


def train_row(model_name, batch, classes, image=448, steps=6, warmup=3):
    torch.manual_seed(0)
    model = bench.build_model(model_name, classes).to(dev).to(memory_format=torch.channels_last)
    model.train()
    crit = torch.nn.CrossEntropyLoss(label_smoothing=0.1)
    opt = torch.optim.SGD(model.parameters(), lr=0.005, momentum=0.9, weight_decay=1e-5)
    images = torch.randn(batch, 3, image, image, device=dev).contiguous(memory_format=torch.channels_last)
    labels = torch.randint(0, classes, (batch,), device=dev)

    def step():
        out = model(images, labels) if model_name == 'APCNN' else model(images)
        loss = sum(crit(o, labels) for o in out[1]) if model_name == 'APCNN' else crit(out, labels)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss
    t_first = time.perf_counter()
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t_warm = time.perf_counter() - t_first
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    rows.append({'model': model_name, 'train_step': f'{image}x{image} batch {batch}, {classes} classes, fp32, channels_last',
                 'ms_per_step': round(ms, 2), 'images_per_sec': round(batch / ms * 1e3, 1), 'warmup_s': round(t_warm, 1),
                 'loss_finite': bool(torch.isfinite(loss).item())})
    del model, opt, images
    torch.cuda.empty_cache()


def mpn_kernels(B=64, d=256, HW=196):
    x = torch.relu(R(B, d, HW)); cov = E(B, d, d); mu = E(B, d); g = R(B, d, d).triu(); dx = E(B, d, HW)
    kernel_row('MPN', 'cov_pool fwd (one kernel: means in LDS + centred Gram)', lambda: lib.hk_cov_pool_fwd(ptr(x), ptr(cov), ptr(mu), B, d, HW, stream()),
               2.0 * B * d * d * HW, 4.0 * B * (d * HW + d * d), flops_exec=2.0 * B * d * d * HW * 10 / 16,
               pmc=('r6_step_MPN_pmc.csv', [('bcnn_gram_panel_kernel<196, 1, true>', 1)]))
    kernel_row('MPN', 'cov_pool bwd', lambda: lib.hk_cov_pool_bwd(ptr(x), ptr(mu), ptr(g), ptr(dx), B, d, HW, stream()),
               2.0 * B * d * d * HW, 4.0 * B * (2 * d * HW + d * d), pmc=('r6_step_MPN_pmc.csv', [('gram_bwd3_kernel<196, 1, 1', 1)]))
    out = E(B, d, d); na = E(B); ys = E(B, 4, d, d); zs = E(B, 4, d, d); da = E(B, d, d)
    nwf = lib.hk_ns_sqrtm_ws_bytes(B, d, 5, 0); nwb = lib.hk_ns_sqrtm_ws_bytes(B, d, 5, 1)
    wsf = E(nwf, dtype=torch.uint8); wsb = E(nwb, dtype=torch.uint8)
    kernel_row('MPN', 'ns_sqrtm fwd chain, general input (12 products of 256^3 per sample, 9 launches x 2 queues)',
               lambda: lib.hk_ns_sqrtm_fwd(ptr(cov), ptr(out), ptr(na), ptr(ys), ptr(zs), B, d, 5, ptr(wsf), nwf, stream()),
               12 * 2.0 * B * d ** 3, 4.0 * B * d * d * 10)
    kernel_row('MPN', 'ns_sqrtm fwd chain, symmetric input = the MPN head (same 12 products, 3 of 4 tiles computed)',
               lambda: lib.hk_ns_sqrtm_fwd_sym(ptr(cov), ptr(out), ptr(na), ptr(ys), ptr(zs), B, d, 5, ptr(wsf), nwf, stream()),
               12 * 2.0 * B * d ** 3, 4.0 * B * d * d * 10, flops_exec=0.75 * 12 * 2.0 * B * d ** 3)
    kernel_row('MPN', 'ns_sqrtm bwd chain (38 products of 256^3 per sample, 13 launches)',
               lambda: lib.hk_ns_sqrtm_bwd(ptr(cov), ptr(out), ptr(na), ptr(ys), ptr(zs), ptr(g), ptr(da), B, d, 5, ptr(wsb),
                                           nwb, stream()), 38 * 2.0 * B * d ** 3, 4.0 * B * d * d * 12,
               flops_exec=34 * 2.0 * B * d ** 3,
               # the backward per step: 2 queues x (8 launches of the general instance + the LAST launch)
               pmc=('r6_step_MPN_pmc.csv', [('nsmm_kernel<64, false, false, false, false>', 16), ('nsmm_kernel<64, false, false, false, true>', 2)]))
    tv = E(B, d * (d + 1) // 2)
    kernel_row('MPN', 'ns_sqrtm fwd chain + triu_vec in its last product (hk_ns_sqrtm_triu_fwd, symmetric: what the MPN head calls)',
               lambda: lib.hk_ns_sqrtm_triu_fwd(ptr(cov), ptr(out), ptr(tv), ptr(na), ptr(ys), ptr(zs), B, d, 5, 1, ptr(wsf), nwf, stream()),
               12 * 2.0 * B * d ** 3, 4.0 * B * d * d * 10, flops_exec=0.75 * 12 * 2.0 * B * d ** 3,
               # the symmetric forward per step: 2 queues x (1 FIRST launch + 8 launches of the plain symmetric instance)
               pmc=('r6_step_MPN_pmc.csv', [('nsmm_kernel<64, false, true, true, false>', 2), ('nsmm_kernel<64, false, true, false, false>', 16)]))
    kernel_row('MPN', 'triu_vec fwd', lambda: lib.hk_triu_vec_fwd(ptr(out), ptr(tv), B, d, stream()), 0, 4.0 * B * 32896 * 2)
    kernel_row('MPN', 'triu_vec bwd', lambda: lib.hk_triu_vec_bwd(ptr(tv), ptr(da), B, d, stream()), 0, 4.0 * B * (32896 + 65536),
               pmc=('r6_step_MPN_pmc.csv', [('triu_bwd_kernel', 1)]))


def cbp_kernels(C=512, HW=196, D=6000):
    plan = F.CbpPlan(*F.sketch_hashes(C, C, D), D, dev)
    for B in (64, 16):
        x = torch.relu(R(B, C, HW)); y = E(B, D); craw = E(B, D); inv = E(B); dy = R(B, D); dx = E(B, C, HW)
        nws = lib.hk_cbp_ws_bytes(B, C, HW, D); ws = E(nws, dtype=torch.uint8)
        fl = 2.0 * B * C * C * HW
        kernel_row('CBCNN', f'cbp fwd B={B} (fused Gram + binning, finish)',
                   lambda: lib.hk_cbp_fwd(ptr(x), ptr(plan.blob), ptr(y), ptr(craw), ptr(inv), B, C, HW, D, ptr(ws), nws, stream()),
                   fl, 4.0 * B * (C * HW + D), flops_exec=fl * 36 / 64,     # Gram tiles J >= I only
                   pmc=('r6_step_CBCNN_bs16_pmc.csv', [('cbp_fused_kernel<196, true>', 1), ('cbp_finish_kernel', 1)]) if B == 16 else None)
        kernel_row('CBCNN', f'cbp bwd B={B} (dc + P generation + GEMM in one kernel)',
                   lambda: lib.hk_cbp_bwd(ptr(x), ptr(plan.blob), ptr(y), ptr(craw), ptr(inv), ptr(dy), ptr(dx), B, C, HW, D,
                                          ptr(ws), nws, stream()), fl, 4.0 * B * (2 * C * HW + 2 * D),
                   pmc=('r6_step_CBCNN_bs16_pmc.csv', [('cbp_bwd3_kernel<196, 1, true, 2>', 1)]) if B == 16 else None)


def apcnn_kernels(B=16, classes=8142):
    for hw in (56, 28, 14):
        n = hw * hw
        ff = R(B, 256, n); aa = torch.rand(B, n, device=dev); gp = E(B, 256); sg = E(B, 256)
        dgp = R(B, 256); dsg = R(B, 256); dff = E(B, 256, n); daa = E(B, n)
        kernel_row('APCNN', f'att_pool fwd {hw}x{hw}', lambda: lib.hk_att_pool_fwd(ptr(ff), ptr(aa), ptr(gp), ptr(sg), B, 256, n, stream()),
                   0, 4.0 * B * 256 * n)
        kernel_row('APCNN', f'att_pool bwd {hw}x{hw}',
                   lambda: lib.hk_att_pool_bwd(ptr(ff), ptr(aa), ptr(dgp), ptr(dsg), ptr(dff), ptr(daa), B, 256, n, stream()),
                   0, 8.0 * B * 256 * n)
    ffs = [R(B, 256, s, s) for s in (56, 28, 14)]
    aas = [torch.rand(B, 1, s, s, device=dev) for s in (56, 28, 14)]
    gp3, sg3, dgp3, dsg3 = E(3, B, 256), E(3, B, 256), R(3, B, 256), R(3, B, 256)
    dfs, das = [torch.empty_like(f) for f in ffs], [torch.empty_like(a) for a in aas]
    tot = sum(4.0 * B * 256 * s * s for s in (56, 28, 14))
    kernel_row('APCNN', 'att_pool3 fwd: the three levels in one launch (what PyramidAttentions calls)',
               lambda: lib.hk_att_pool3_fwd(ptr(ffs[0]), ptr(ffs[1]), ptr(ffs[2]), ptr(aas[0]), ptr(aas[1]), ptr(aas[2]), ptr(gp3), ptr(sg3),
                                            B, 256, 3136, 784, 196, stream()), 0, tot,
               pmc=('r6_step_APCNN_8142_pmc.csv', [('att_pool3_fwd_kernel', 1)]))
    kernel_row('APCNN', 'att_pool3 bwd: the three levels in one launch',
               lambda: lib.hk_att_pool3_bwd(ptr(ffs[0]), ptr(ffs[1]), ptr(ffs[2]), ptr(aas[0]), ptr(aas[1]), ptr(aas[2]), ptr(dgp3), ptr(dsg3),
                                            ptr(dfs[0]), ptr(dfs[1]), ptr(dfs[2]), ptr(das[0]), ptr(das[1]), ptr(das[2]),
                                            B, 256, 3136, 784, 196, stream()), 0, 2 * tot,
               pmc=('r6_step_APCNN_8142_pmc.csv', [('att_pool3_bwd_kernel', 1)]))
    masks = [torch.rand(B, 1, s, s, device=dev) for s in (56, 28, 14)]
    lv = [(8, 64., 5), (16, 128., 3), (32, 256., 1)]
    tabs = []

    def roi_all():                                     # border kept for != 200 classes: [0.1 h, 0.9 h)  (APCNN.py:451-454)
        tabs.clear()
        for m, (s, a, k) in zip(masks, lv):
            tabs.append(F.att_roi_select(m, s, a, 448, 448, classes, 0.05, k))
    kernel_row('APCNN', f'att_roi_select, three launches ({classes}-class border; round 2)', roi_all)

    def roi_one():                                     # what the AP-CNN forward calls: one launch, grid B x 3
        tabs[:] = F.att_roi_select_levels(masks, lv, 448, 448, classes, 0.05)
    kernel_row('APCNN', f'att_roi_select3: the three levels in one launch ({classes}-class border)', roi_one,
               pmc=('r6_step_APCNN_8142_pmc.csv', [('att_roi_select3_kernel', 1)]))
    u = torch.rand(B, 2, device=dev)
    box, drop = F.roi_boxes(tabs, u, 8.0)
    x2 = R(B, 512, 56, 56); y2 = E(B, 512, 56, 56)
    kernel_row('APCNN', 'roi_crop_resize fwd', lambda: lib.hk_roi_crop_resize_fwd(ptr(x2), ptr(box), ptr(drop), ptr(y2), B, 512, 56, 56, 1, stream()),
               0, 8.0 * B * 512 * 3136, pmc=('r6_step_APCNN_8142_pmc.csv', [('roi_crop_fwd_tab2_kernel', 1)]))
    kernel_row('APCNN', 'roi_crop_resize bwd', lambda: lib.hk_roi_crop_resize_bwd(ptr(y2), ptr(box), ptr(drop), ptr(x2), B, 512, 56, 56, 1, stream()),
               0, 8.0 * B * 512 * 3136, pmc=('r6_step_APCNN_8142_pmc.csv', [('roi_crop_bwd_tab3_kernel', 1)]))


def ssqrt_kernels(B=64, C=512, HW=196):
    """The signed-sqrt variant (BCNN.py:23-24) with and without the l2 scale folded into the classifier."""
    x = torch.randn(B, C, HW, device=dev); y = E(B, C * C); inv = E(B); dy = R(B, C * C); dx = E(B, C, HW)
    nws = lib.hk_bcnn_ssqrt_ws_bytes(B, C, HW); ws = E(nws, dtype=torch.uint8)
    fl, by = 2.0 * B * C * C * HW, 4.0 * B * (C * HW + C * C)
    kernel_row('BCNN-ssqrt', 'ssqrt pool fwd (Gram with sign-sqrt epilogue + scale pass)',
               lambda: lib.hk_bcnn_ssqrt_pool_fwd(ptr(x), ptr(y), ptr(inv), B, C, HW, ptr(ws), nws, stream()), fl, by, flops_exec=fl * 36 / 64)
    kernel_row('BCNN-ssqrt', 'ssqrt pool fwd, unscaled (the 1 / |z| goes into the classifier epilogue: no scale pass)',
               lambda: lib.hk_bcnn_ssqrt_pool_fwd_unscaled(ptr(x), ptr(y), ptr(inv), B, C, HW, ptr(ws), nws, stream()), fl, by,
               flops_exec=fl * 36 / 64)
    kernel_row('BCNN-ssqrt', 'ssqrt pool bwd', lambda: lib.hk_bcnn_ssqrt_pool_bwd(ptr(x), ptr(y), ptr(dy), ptr(inv), ptr(dx), B, C, HW, ptr(ws), nws, stream()),
               fl, 4.0 * B * (2 * C * HW + 2 * C * C))
    kernel_row('BCNN-ssqrt', 'ssqrt pool bwd, unscaled', lambda: lib.hk_bcnn_ssqrt_pool_bwd_unscaled(ptr(x), ptr(y), ptr(dy), ptr(inv), ptr(dx), B, C, HW, ptr(ws), nws, stream()),
               fl, 4.0 * B * (2 * C * HW + 2 * C * C))
    import ctypes
    w = R(200, C * C); out = E(B, 200); bi0 = R(200)
    nwl = lib.hk_linear_ws_bytes(B, C * C, 200); wsl = E(nwl, dtype=torch.uint8)
    npart = ctypes.c_int(0)

    def pair_norm():
        lib.hk_bcnn_ssqrt_pool_fwd_unscaled(ptr(x), ptr(y), ptr(inv), B, C, HW, ptr(ws), nws, stream())
        return lib.hk_linear_fwd_scaled(ptr(y), ptr(w), ptr(bi0), ptr(inv), ptr(out), B, C * C, 200, ptr(wsl), nwl, stream())

    def pair_parts():
        lib.hk_bcnn_ssqrt_pool_fwd_parts(ptr(x), ptr(y), ptr(ws), ctypes.byref(npart), B, C, HW, stream())
        return lib.hk_linear_fwd_ssq(ptr(y), ptr(w), ptr(bi0), ptr(ws), npart.value, ptr(inv), ptr(out), B, C * C, 200, ptr(wsl), nwl,
                                     stream())
    flp = fl + 2.0 * B * C * C * 200
    byp = by + 4.0 * (B * C * C + 200 * C * C)
    kernel_row('BCNN-ssqrt', 'ssqrt pool fwd unscaled + classifier fwd (4 launches: Gram, norm, GEMM, reduce)', pair_norm, flp, byp, flops_exec=flp - fl * 28 / 64)
    kernel_row('BCNN-ssqrt', 'ssqrt pool fwd parts + classifier fwd (3 launches: the norm formed in the reduce launch: what the fused node runs)',
               pair_parts, flp, byp, flops_exec=flp - fl * 28 / 64)
    g, lo, bi = R(B, 200), R(B, 200), R(200)
    kernel_row('BCNN-ssqrt', 'ssqrt pool bwd, unscaled, <y, dy> handed over by the classifier (hk_bcnn_ssqrt_pool_bwd_tdot: what the fused node runs)',
               lambda: lib.hk_bcnn_ssqrt_pool_bwd_tdot(ptr(x), ptr(y), ptr(dy), ptr(inv), ptr(g), ptr(lo), ptr(bi), 200, 1, ptr(dx), B, C, HW,
                                                       ptr(ws), nws, stream()),
               fl, 4.0 * B * (2 * C * HW + 2 * C * C))


def guarded(fn, *args):
    try:
        fn(*args)
    except Exception as e:  # noqa: BLE001
        rows.append({'stage': getattr(fn, '__name__', str(fn)) + str(args), 'error': repr(e)[:300]})
        torch.cuda.empty_cache()


if __name__ == '__main__':
    guarded(ssqrt_kernels)
    if '--ssqrt-only' in sys.argv:
        print(json.dumps(rows), flush=True)
        sys.exit(0)
    guarded(mpn_kernels)
    guarded(cbp_kernels)
    guarded(apcnn_kernels)
    if '--kernels-only' in sys.argv:
        print(json.dumps(rows), flush=True)
        sys.exit(0)
    steps = 3 if QUICK else 6
    guarded(train_row, 'MPN', 64, 200, 448, steps)
    guarded(train_row, 'CBCNN', 64, 200, 448, steps)
    guarded(train_row, 'CBCNN', 16, 200, 448, steps)
    guarded(train_row, 'APCNN', 16, 8142, 448, steps)
    print(json.dumps(rows), flush=True)
