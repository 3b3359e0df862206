"""Where a K-block of the Gram backward goes: cycle stamps (s_memtime) taken by wave 0 of the first 64 workgroups at the
phase boundaries of every K-block, from the instrumented build (make -C hawkeye_amd/csrc lab).  Prints the mean cycles
per phase for the 64-row kernel (bwd_v=1) and the 128-row kernel (bwd_v=5) at B = 64, C = 512, 14x14.
    python tools/bwd_lab.py"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

lib = ctypes.CDLL(os.path.join(ROOT, 'hawkeye_amd', 'csrc', 'libhawkeye_hip_lab.so'))
P, I = ctypes.c_void_p, ctypes.c_int
lib.hk_bcnn_colsum_norm.argtypes = [P, P, P, I, I, I, P, ctypes.c_size_t, P]
lib.hk_bcnn_gram_norm.argtypes = [P, P, P, I, I, I, P]
lib.hk_bcnn_bwd_gemm.argtypes = [P, P, P, P, P, P, I, I, I, P]
lib.hk_bcnn_pool_ws_bytes.restype = ctypes.c_size_t
lib.hk_bcnn_pool_ws_bytes.argtypes = [I, I, I]
lib.hk_tuning_set.argtypes = [ctypes.c_char_p, I]
lib.hk_lab_set_stamps.argtypes = [P]

dev = torch.device('cuda:0')
B, C, HW = 64, 512, 196
x = torch.relu(torch.randn(B, C, HW, device=dev))
y, dy, dx = torch.empty(B, C * C, device=dev), torch.randn(B, C * C, device=dev), torch.empty_like(x)
inv, cs, tp = torch.empty(B, device=dev), torch.empty(B, HW, device=dev), torch.empty(B, C // 64, device=dev)
nws = lib.hk_bcnn_pool_ws_bytes(B, C, HW)
ws = torch.empty(nws, dtype=torch.uint8, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def p(t):
    return ctypes.c_void_p(t.data_ptr())


lib.hk_bcnn_colsum_norm(p(x), p(cs), p(inv), B, C, HW, p(ws), nws, st)
lib.hk_bcnn_gram_norm(p(x), p(inv), p(y), B, C, HW, st)
stamps = torch.zeros(64, 32, 8, dtype=torch.int64, device=dev)
KERNELS = (
    (1, ['scatter dy^T + barrier', 'P tile + barrier', 'X block + barrier', 'issue next loads', 'MFMA phase'], 8),
    (5, ['first fragments', 'MFMA groups 0-3 + loads of the next block', 'MFMA groups 4-7 + LDS stores', 'barrier'], 16),
)
out = {}
for v, names, nkb in KERNELS:
    lib.hk_tuning_set(b'bwd_v', v)
    lib.hk_lab_set_stamps(None)
    for _ in range(3):
        lib.hk_bcnn_bwd_gemm(p(x), p(y), p(dy), p(inv), p(dx), p(tp), B, C, HW, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.hk_bcnn_bwd_gemm(p(x), p(y), p(dy), p(inv), p(dx), p(tp), B, C, HW, st)
    e1.record()
    torch.cuda.synchronize()
    us_plain = e0.elapsed_time(e1) / 20 * 1e3
    stamps.zero_()
    lib.hk_lab_set_stamps(p(stamps))
    lib.hk_bcnn_bwd_gemm(p(x), p(y), p(dy), p(inv), p(dx), p(tp), B, C, HW, st)
    torch.cuda.synchronize()
    s = stamps.cpu()[:, :nkb, :].double()
    nslot = len(names) + 1
    d = s[:, :, 1:nslot] - s[:, :, :nslot - 1]                                   # phase durations inside a K-block
    rows = {n: round(float(d[:, 1:nkb - 1, i].mean())) for i, n in enumerate(names)}   # steady state: skip first / last
    rows['to the next block top'] = round(float((s[:, 2:nkb, 0] - s[:, 1:nkb - 1, nslot - 1]).mean()))
    total = float((s[:, 2:nkb, 0] - s[:, 1:nkb - 1, 0]).mean())
    span = float((s[:, nkb - 1, nslot - 1] - s[:, 0, 0]).mean())
    out[f'bwd_v={v}'] = {'us_uninstrumented': round(us_plain, 1), 'cycles_per_kblock': round(total), 'phases': rows,
                         'first_to_last_stamp_cycles': round(span)}
lib.hk_lab_set_stamps(None)
lib.hk_tuning_set(b'bwd_v', 0)
print(json.dumps(out, indent=1))
