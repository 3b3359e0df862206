"""Timing-only variants of the 128-row Gram backward (instrumented build): which part of a K-block costs what."""
import ctypes, json, os, sys
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
dev = torch.device('cuda:0')
B, C, HW = 64, 512, 196
x = torch.relu(torch.randn(B, C, HW, device=dev))
y, dy, dx = torch.empty(B, C * C, device=dev), torch.randn(B, C * C, device=dev), torch.empty_like(x)
inv, cs, tp = torch.empty(B, device=dev), torch.empty(B, HW, device=dev), torch.empty(B, C // 64, device=dev)
nws = lib.hk_bcnn_pool_ws_bytes(B, C, HW)
ws = torch.empty(nws, dtype=torch.uint8, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr())
lib.hk_bcnn_colsum_norm(p(x), p(cs), p(inv), B, C, HW, p(ws), nws, st)
lib.hk_bcnn_gram_norm(p(x), p(inv), p(y), B, C, HW, st)
out = {}
for rnd in range(2):
    for v, tag in ((1, '64-row kernel'), (5, '128-row kernel'), (6, '128-row, no staging in the loop'), (7, '128-row, fragments read once per K-block'), (8, '128-row, neither'),
                   (9, '128-row staged by LDS-DMA'), (10, '128-row LDS-DMA, every K-block staged from the addresses of K-block 0 (L2 hits)')):
        lib.hk_tuning_set(b'bwd_v', v)
        for _ in range(3):
            lib.hk_bcnn_bwd_gemm(p(x), p(y), p(dy), p(inv), p(dx), p(tp), B, C, HW, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            lib.hk_bcnn_bwd_gemm(p(x), p(y), p(dy), p(inv), p(dx), p(tp), B, C, HW, st)
        e1.record()
        torch.cuda.synchronize()
        out.setdefault(tag, []).append(round(e0.elapsed_time(e1) / 30 * 1e3, 1))
lib.hk_tuning_set(b'bwd_v', 0)
print(json.dumps(out, indent=1))
