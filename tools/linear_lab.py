"""Which side of the wide-classifier forward (hk_linear_fwd at 64 x 262144 -> 200) the time belongs to: the instrumented
build (make -C hawkeye_amd/csrc lab) runs linear_skinny_kernel without its MFMAs (the LDS-DMA stream alone) and without
the LDS-DMA inside the loop (MFMAs + fragment reads alone), without the fragment reads (MFMAs + LDS-DMA), and with the
MFMAs and barriers alone.      python tools/linear_lab.py"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

lib = ctypes.CDLL(os.path.join(ROOT, 'hawkeye_amd', 'csrc', 'libhawkeye_hip_lab.so'))
P, I, SZ = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
lib.hk_linear_ws_bytes.restype = SZ; lib.hk_linear_ws_bytes.argtypes = [I, I, I]
lib.hk_linear_fwd.argtypes = [P, P, P, P, I, I, I, P, SZ, P]
lib.hk_lab_set_linear_mode.argtypes = [I]
dev = torch.device('cuda:0')
p = lambda t: P(t.data_ptr())
st = P(torch.cuda.current_stream().cuda_stream)
out = {}
for tag, B, J, K in (('bcnn 64 x 262144 -> 200', 64, 262144, 200), ('osme 10 x 100352 -> 1024', 10, 100352, 1024)):
    y, w, bias = torch.randn(B, J, device=dev), torch.randn(K, J, device=dev) * 0.01, torch.randn(K, device=dev)
    o = torch.empty(B, K, device=dev)
    nws = lib.hk_linear_ws_bytes(B, J, K)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    fw = lambda: lib.hk_linear_fwd(p(y), p(w), p(bias), p(o), B, J, K, p(ws), nws, st)
    res = {}
    for rnd in range(3):
        for mode, name in ((0, 'kernel as shipped'), (1, 'no MFMAs (LDS-DMA stream + fragment reads)'), (2, 'no LDS-DMA in the loop (MFMAs + fragment reads)'),
                           (3, 'neither (fragment reads + barriers)'), (4, 'no fragment reads (MFMAs + LDS-DMA)'),
                           (6, 'MFMAs + barriers alone')):
            assert lib.hk_lab_set_linear_mode(mode) == 0
            for _ in range(3):
                assert fw() == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fw()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(name, []).append(round(e0.elapsed_time(e1) * 50.0, 2))
    lib.hk_lab_set_linear_mode(0)
    out[tag] = {k: sorted(v)[1] for k, v in res.items()}
    del y, w
print(json.dumps(out, indent=1))
