"""The shader clock the chip runs at while each hot-path kernel executes (tools/probe/clk_sampler.hip: one wave samples the
cycle counter against the 100 MHz reference every 5 us on a second HIP queue while the kernel under test is launched
back to back for ~4 ms).  The datasheet's 157.3 TF/s is 256 CU x 256 FLOP/cycle x 2.4 GHz; `mfma_peak_at_clock` is the
same product at the clock the power manager actually grants that kernel - the part of `1 - frac` no kernel structure can
win back.      make -C tools/probe && python tools/clock_probe.py > gpurun_out/clock_probe.json"""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hawkeye_amd import _lib
from hawkeye_amd._lib import ptr, stream

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
probe = ctypes.CDLL(os.path.join(ROOT, 'tools', 'probe', 'libhk_probe.so'))
probe.hk_probe_clk_sampler.restype = ctypes.c_int
probe.hk_probe_clk_sampler.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
probe.hk_probe_mfma.restype = ctypes.c_longlong
probe.hk_probe_mfma.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
lib = _lib.load()
dev = torch.device('cuda:0')
side = torch.cuda.Stream()

N, PERIOD = 1000, 500                  # 1000 samples, 5 us apart = 5 ms
samples = torch.zeros(3 * N, dtype=torch.int64, device=dev)


def measure(name, fn, flops=None, window_ms=4.0):
    """fn() enqueues one launch on the current stream."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    reps = max(3, int(window_ms * 1e3 / us))
    samples.zero_()
    torch.cuda.synchronize()
    assert probe.hk_probe_clk_sampler(samples.data_ptr(), N, PERIOD, ctypes.c_void_p(side.cuda_stream)) == 0
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    a = samples.cpu().numpy().astype(np.float64).reshape(N, 3)
    a = a[(a[:, 2] - a[:, 1]) <= 40]                          # both reference reads within 0.4 us of each other
    cyc, rt = a[:, 0], 0.5 * (a[:, 1] + a[:, 2])
    t = (rt - rt[0]) * 0.01                                   # us since the first sample
    end = min(reps * us, N * PERIOD * 0.01)

    def clock(lo, hi):                                        # cycles / time over [lo, hi] us
        k = np.nonzero((t >= lo) & (t <= hi))[0]
        return float((cyc[k[-1]] - cyc[k[0]]) / ((rt[k[-1]] - rt[k[0]]) * 0.01)) if len(k) > 4 else None

    parts = [clock(lo, lo + 100.0) for lo in np.arange(300.0, end - 300.0, 100.0)]
    parts = [p for p in parts if p]
    busy = float(np.median(parts)) if parts else None           # median of the 100-us windows while the kernels run
    row = {'kernel': name, 'us_per_launch': round(us, 1), 'launches_in_window': reps,
           'clock_mhz': round(busy, 0) if busy else None,
           'clock_mhz_min_100us': round(min(parts), 0) if parts else None,
           'clock_mhz_max_100us': round(max(parts), 0) if parts else None,
           'clock_mhz_idle_after': (lambda c: round(c, 0) if c else None)(clock(end + 300.0, N * PERIOD * 0.01))}
    if row['clock_mhz']:
        row['mfma_peak_at_clock_tflops'] = round(256 * 256 * row['clock_mhz'] * 1e6 / 1e12, 1)
        if flops:
            tf = flops / us / 1e6
            row.update(tflops_executed=round(tf, 1), frac_of_157_3=round(tf / 157.3, 3),
                       frac_at_clock=round(tf / row['mfma_peak_at_clock_tflops'], 3))
    print(json.dumps(row), file=sys.stderr, flush=True)
    return row


rows = []
st = stream
# nothing but MFMAs (32x32x2, 8 waves per CU)
src, out = torch.randn(65536, device=dev), torch.empty(512 * 512, device=dev)
nm = [0]
for kind, nm_, fl in ((2, 'v_mfma_f32_32x32x2_f32', 4096.0), (3, 'v_mfma_f32_32x32x2_f32, operands changing every instruction', 4096.0),
                      (0, 'v_mfma_f32_16x16x4_f32', 2048.0)):
    def mfma_only():
        nm[0] = probe.hk_probe_mfma(src.data_ptr(), out.data_ptr(), kind, 256, 512, 2000 if kind >= 2 else 1000,
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    mfma_only()
    rows.append(measure(f'MFMA only: {nm_} back to back, 8 waves per CU', mfma_only, flops=nm[0] * fl * 256 * 8))
for thr, mode, tiles, what in ((256, 0, 400, 'nothing else'), (512, 0, 400, 'nothing else'),
                               (256, 1, 400, '+ a workgroup barrier per tile (24 steps)'),
                               (256, 3, 400, '+ barrier + the epilogue arithmetic (sum, sqrt, fma; accumulators restart)'),
                               (256, 7, 400, '+ barrier + epilogue + 64 B per lane stored per tile'),
                               (256, 7, 40, '+ barrier + epilogue + stores, 40 tiles per launch'),
                               (256, 7, 9, '+ barrier + epilogue + stores, 9 tiles per launch (the Gram forward\'s length)'),
                               (256, 0, 9, 'nothing else, 9 tiles per launch'),
                               (256, 9, 400, '+ barrier + the next panel staged (13 x 16 B loads per thread from L2, written to LDS inside the MFMA steps)'),
                               (256, 15, 400, '+ barrier + epilogue + stores + panel staging'),
                               (256, 15, 9, '+ barrier + epilogue + stores + panel staging, 9 tiles per launch')):
    def mfma_lds():
        nm[0] = probe.hk_probe_mfma(src.data_ptr(), out.data_ptr(), 4 + mode, 256, thr, 24 * tiles, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    mfma_lds()
    rows.append(measure(f'MFMA fed from LDS (32x32x2, one ds_read_b128 per two MFMAs: the Gram kernels\' diet), {thr // 64} waves per CU, {what}',
                        mfma_lds, flops=nm[0] * 4096.0 * 256 * (thr // 64)))
for tiles in (400, 9):
    def mfma_lds8():
        nm[0] = probe.hk_probe_mfma(src.data_ptr(), out.data_ptr(), 30, 256, 512, 24 * tiles, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    mfma_lds8()
    rows.append(measure(f'MFMA fed from LDS, EIGHT waves (K split between the two waves of a SIMD, partial sums exchanged through LDS), barrier + epilogue + stores + panel staging, {tiles} tiles per launch',
                        mfma_lds8, flops=nm[0] * 4096.0 * 256 * 8))
# an HBM copy
big_a, big_b = torch.empty(1 << 28, device=dev), torch.empty(1 << 28, device=dev)
rows.append(measure('HBM copy 1 GiB -> 1 GiB (torch copy_)', lambda: big_b.copy_(big_a)))
del big_a, big_b

# BCNN pooling head, B = 64, C = 512, 14 x 14
B, C, HW = 64, 512, 196
x = torch.relu(torch.randn(B, C, HW, device=dev)); y = torch.empty(B, C * C, device=dev); dy = torch.randn(B, C * C, device=dev)
dx = torch.empty_like(x); inv = torch.rand(B, device=dev) + 0.5; tp = torch.empty(B, C // 64, device=dev)
rows.append(measure('Gram forward (bcnn_gram_*_kernel<196>)', lambda: lib.hk_bcnn_gram_norm(ptr(x), ptr(inv), ptr(y), B, C, HW, st()),
                    flops=2.0 * B * C * C * HW * 36 / 64))
lib.hk_bcnn_gram_norm(ptr(x), ptr(inv), ptr(y), B, C, HW, st())
rows.append(measure('Gram backward (gram_bwd3_kernel<196,0,2>)',
                    lambda: lib.hk_bcnn_bwd_gemm(ptr(x), ptr(y), ptr(dy), ptr(inv), ptr(dx), ptr(tp), B, C, HW, st()), flops=2.0 * B * C * C * HW))
del y, dy

# classifier 262144 -> 200
Bl, J, K = 64, 262144, 200
yl, wl, bl = torch.randn(Bl, J, device=dev), torch.randn(K, J, device=dev) * 0.01, torch.zeros(K, device=dev)
g, o = torch.randn(Bl, K, device=dev), torch.empty(Bl, K, device=dev)
dyl, dwl, dbl = torch.empty(Bl, J, device=dev), torch.empty(K, J, device=dev), torch.empty(K, device=dev)
nws = lib.hk_linear_ws_bytes(Bl, J, K); ws = torch.empty(nws, dtype=torch.uint8, device=dev)
rows.append(measure('classifier forward (linear_skinny_kernel<13,4> + reduce)',
                    lambda: lib.hk_linear_fwd(ptr(yl), ptr(wl), ptr(bl), ptr(o), Bl, J, K, ptr(ws), nws, st()), flops=2.0 * Bl * J * K))
# the same forward kernel taken apart (tools/probe/linear_lab.hip): where the clock goes when the halves run alone
lab = ctypes.CDLL(os.path.join(ROOT, 'tools', 'probe', 'libhk_linear_lab.so'))
P = ctypes.c_void_p
lab.hk_probe_linear_fwd.argtypes = [ctypes.c_int, P, P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, P]
part = torch.empty(256 * Bl * K, device=dev)
for m, nm_ in ((0, 'whole kernel'), (2, 'no loads in the loop: MFMAs + fragment reads + barriers'), (9, 'LDS-DMA stream only: no MFMAs, no fragment reads')):
    rows.append(measure('classifier forward, lab build - ' + nm_,
                        lambda: lab.hk_probe_linear_fwd(m, ptr(yl), ptr(wl), ptr(part), Bl, J, K, 1, st()),
                        flops=2.0 * Bl * J * K if m != 9 else None))
rows.append(measure('classifier backward (linear_bwd64_kernel<50,0>)',
                    lambda: lib.hk_linear_bwd(ptr(yl), ptr(wl), ptr(g), ptr(dyl), ptr(dwl), ptr(dbl), Bl, J, K, st()), flops=4.0 * Bl * J * K))
del yl, wl, dyl, dwl

# Newton-Schulz forward chain (B = 64, d = 256, 5 iterations, symmetric input)
import hawkeye_amd.functional as F
xc = torch.relu(torch.randn(64, 256, 14, 14, device=dev))
cov = F.covpool(xc)
rows.append(measure('Newton-Schulz forward chain (9 nsmm_kernel launches, two queues)', lambda: F.sqrtm(cov, 5, symmetric=True),
                    flops=9 * 2.0 * 256 ** 3 * 64 * 0.75))
json.dump(rows, sys.stdout, indent=0)
