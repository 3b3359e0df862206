"""A/B timing of the Newton-Schulz chain (B = 64, d = 256, iterN = 5) over the dispatch knobs, several rounds with the
configurations interleaved (the first-measured configuration otherwise reads 3-7 % slower: clocks, caches):
    python tools/ns_bench.py [rounds]      -> one JSON list: median fwd / bwd us per configuration"""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hawkeye_amd import _lib
from hawkeye_amd._lib import ptr, stream

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B, d, HW = 64, 256, 196
lib = _lib.load()
dev = torch.device('cuda:0')
x = torch.relu(torch.randn(B, d, HW, device=dev))
cov, mu = torch.empty(B, d, d, device=dev), torch.empty(B, d, device=dev)
lib.hk_cov_pool_fwd(ptr(x), ptr(cov), ptr(mu), B, d, HW, stream())
out, na = torch.empty(B, d, d, device=dev), torch.empty(B, device=dev)
ys, zs = torch.empty(B, 4, d, d, device=dev), torch.empty(B, 4, d, d, device=dev)
g, da = torch.randn(B, d, d, device=dev).triu(), torch.empty(B, d, d, device=dev)
nwf, nwb = lib.hk_ns_sqrtm_ws_bytes(B, d, 5, 0), lib.hk_ns_sqrtm_ws_bytes(B, d, 5, 1)
wf, wb = torch.empty(nwf, dtype=torch.uint8, device=dev), torch.empty(nwb, dtype=torch.uint8, device=dev)


def fwd():
    return lib.hk_ns_sqrtm_fwd(ptr(cov), ptr(out), ptr(na), ptr(ys), ptr(zs), B, d, 5, ptr(wf), nwf, stream())


def bwd():
    return lib.hk_ns_sqrtm_bwd(ptr(cov), ptr(out), ptr(na), ptr(ys), ptr(zs), ptr(g), ptr(da), B, d, 5, ptr(wb), nwb, stream())


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        assert fn() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


configs = [(tn, ns) for ns in (0, 1) for tn in (0, 64, 128)]
res = {c: ([], []) for c in configs}
ref = None
for r in range(rounds):
    for c in (configs if r % 2 == 0 else configs[::-1]):
        lib.hk_tuning_set(b'ns_tn', c[0])
        lib.hk_tuning_set(b'ns_streams', c[1])
        res[c][0].append(timeit(fwd))
        res[c][1].append(timeit(bwd))
        torch.cuda.synchronize()
        if ref is None:
            ref = (out.clone(), da.clone())
        else:
            assert torch.equal(out, ref[0]) and torch.equal(da, ref[1]), c
lib.hk_tuning_set(b'ns_tn', 0)
lib.hk_tuning_set(b'ns_streams', 0)
rows = []
for c in configs:
    f, b = statistics.median(res[c][0]), statistics.median(res[c][1])
    rows.append({'ns_tn': c[0], 'ns_streams': c[1], 'fwd_us': round(f, 1), 'bwd_us': round(b, 1),
                 'fwd_frac': round(12 * 2.0 * B * d ** 3 / f / 1e6 / 157.3, 3), 'bwd_frac': round(38 * 2.0 * B * d ** 3 / b / 1e6 / 157.3, 3),
                 'fwd_all': [round(v, 1) for v in res[c][0]], 'bwd_all': [round(v, 1) for v in res[c][1]]})
print(json.dumps(rows))
