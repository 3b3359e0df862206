"""Socket power and clocks as rocm-smi reports them while one hot-path kernel is launched back to back for ~3 s - the
slow-sensor companion of tools/clock_probe.py (which reads the shader clock from inside the GPU at 5-us resolution).
    make -C tools/probe && python tools/power_probe.py > gpurun_out/power_probe.json"""
import ctypes
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hawkeye_amd import _lib
from hawkeye_amd._lib import ptr, stream

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
probe = ctypes.CDLL(os.path.join(ROOT, 'tools', 'probe', 'libhk_probe.so'))
probe.hk_probe_mfma.restype = ctypes.c_longlong
probe.hk_probe_mfma.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
lib = _lib.load()
dev = torch.device('cuda:0')


def smi():
    try:
        out = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--json'], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(out)
        card = d[sorted(d)[0]]
        keep = {k: v for k, v in card.items() if any(t in k.lower() for t in ('power', 'sclk', 'mclk', 'fclk'))}
        return keep
    except Exception as e:              # noqa: BLE001 - a probe: report, do not die
        return {'error': repr(e)}


def run(name, fn, seconds=3.0):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    samples, stop = [], threading.Event()

    def poll():
        time.sleep(0.8)
        while not stop.is_set():
            samples.append(smi())
            time.sleep(0.4)

    th = threading.Thread(target=poll)
    th.start()
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        for _ in range(200):
            fn()
        n += 200
        torch.cuda.synchronize()
    el = time.time() - t0
    stop.set()
    th.join()
    row = {'kernel': name, 'launches': n, 'us_per_launch_incl_host': round(el / n * 1e6, 1), 'rocm_smi': samples}
    print(json.dumps(row)[:600], file=sys.stderr, flush=True)
    return row


rows = [{'kernel': 'idle', 'rocm_smi': [smi()]}]
src, out = torch.randn(65536, device=dev), torch.empty(512 * 512, device=dev)
rows.append(run('MFMA only (32x32x2, 8 waves per CU)',
                lambda: probe.hk_probe_mfma(src.data_ptr(), out.data_ptr(), 2, 256, 512, 400, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))))
rows.append(run('MFMA only (32x32x2, operands changing every instruction)',
                lambda: probe.hk_probe_mfma(src.data_ptr(), out.data_ptr(), 3, 256, 512, 400, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))))
rows.append(run('MFMA only (16x16x4, 8 accumulators, 8 waves per CU)',
                lambda: probe.hk_probe_mfma(src.data_ptr(), out.data_ptr(), 0, 256, 512, 200, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))))
B, C, HW = 64, 512, 196
x = torch.relu(torch.randn(B, C, HW, device=dev)); y = torch.empty(B, C * C, device=dev); dy = torch.randn(B, C * C, device=dev)
dx = torch.empty_like(x); inv = torch.rand(B, device=dev) + 0.5; tp = torch.empty(B, C // 64, device=dev)
rows.append(run('Gram forward', lambda: lib.hk_bcnn_gram_norm(ptr(x), ptr(inv), ptr(y), B, C, HW, stream())))
rows.append(run('Gram backward', lambda: lib.hk_bcnn_bwd_gemm(ptr(x), ptr(y), ptr(dy), ptr(inv), ptr(dx), ptr(tp), B, C, HW, stream())))
del y, dy
Bl, J, K = 64, 262144, 200
yl, wl, bl = torch.randn(Bl, J, device=dev), torch.randn(K, J, device=dev) * 0.01, torch.zeros(K, device=dev)
g, o = torch.randn(Bl, K, device=dev), torch.empty(Bl, K, device=dev)
dyl, dwl, dbl = torch.empty(Bl, J, device=dev), torch.empty(K, J, device=dev), torch.empty(K, device=dev)
nws = lib.hk_linear_ws_bytes(Bl, J, K); ws = torch.empty(nws, dtype=torch.uint8, device=dev)
rows.append(run('classifier forward', lambda: lib.hk_linear_fwd(ptr(yl), ptr(wl), ptr(bl), ptr(o), Bl, J, K, ptr(ws), nws, stream())))
rows.append(run('classifier backward', lambda: lib.hk_linear_bwd(ptr(yl), ptr(wl), ptr(g), ptr(dyl), ptr(dwl), ptr(dbl), Bl, J, K, stream())))
big_a, big_b = torch.empty(1 << 28, device=dev), torch.empty(1 << 28, device=dev)
rows.append(run('HBM copy 1 GiB', lambda: big_b.copy_(big_a)))
json.dump(rows, sys.stdout, indent=0)
