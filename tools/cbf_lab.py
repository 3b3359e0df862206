"""Where a tile of the fused CBP forward goes: cycle stamps (s_memtime) of an MFMA wave and a binning wave of the first
64 workgroups, from the instrumented build (make -C hawkeye_amd/csrc lab).      python tools/cbf_lab.py [B=64]"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import hawkeye_amd.functional as F

lib = ctypes.CDLL(os.path.join(ROOT, 'hawkeye_amd', 'csrc', 'libhawkeye_hip_lab.so'))
P, I, SZ = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
lib.hk_cbp_plan_bytes.restype = SZ; lib.hk_cbp_plan_bytes.argtypes = [I, I]
lib.hk_cbp_plan_build.argtypes = [P, P, P, P, I, I, P, P]
lib.hk_cbp_ws_bytes.restype = SZ; lib.hk_cbp_ws_bytes.argtypes = [I, I, I, I]
lib.hk_cbp_fwd.argtypes = [P, P, P, P, P, I, I, I, I, P, SZ, P]
lib.hk_tuning_set.argtypes = [ctypes.c_char_p, I]
lib.hk_lab_set_cbf_stamps.argtypes = [P]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device('cuda:0')
C, HW, D = 512, 196, 6000
p = lambda t: P(t.data_ptr())
st = P(torch.cuda.current_stream().cuda_stream)
h1, s1, h2, s2 = [np.ascontiguousarray(a) for a in F.sketch_hashes(C, C, D)]
blob = torch.empty(lib.hk_cbp_plan_bytes(C, D), dtype=torch.uint8, device=dev)
assert lib.hk_cbp_plan_build(h1.ctypes.data, s1.ctypes.data, h2.ctypes.data, s2.ctypes.data, C, D, p(blob), st) == 0
x = torch.relu(torch.randn(B, C, HW, device=dev))
y, cr, inv = torch.empty(B, D, device=dev), torch.empty(B, D, device=dev), torch.empty(B, device=dev)
nws = lib.hk_cbp_ws_bytes(B, C, HW, D)
ws = torch.empty(nws, dtype=torch.uint8, device=dev)
fw = lambda: lib.hk_cbp_fwd(p(x), p(blob), p(y), p(cr), p(inv), B, C, HW, D, p(ws), nws, st)
stamps = torch.zeros(64, 32, 16, dtype=torch.int64, device=dev)
out = {}
for v in (3, 4):
    lib.hk_tuning_set(b'cbp_bin', v)
    lib.hk_lab_set_cbf_stamps(None)
    for _ in range(3):
        assert fw() == 0
    stamps.zero_()
    lib.hk_lab_set_cbf_stamps(p(stamps))
    fw()
    torch.cuda.synchronize()
    s = stamps.cpu().double()
    nt = int((s[0, :, 0] > 0).sum())
    mid = slice(1, max(nt - 1, 2))
    r = {'tiles': nt}
    r['M loop (stamp0->1)'] = round(float((s[:, mid, 1] - s[:, mid, 0]).mean()))
    r['M T-store + wait B1 (1->2)'] = round(float((s[:, mid, 2] - s[:, mid, 1]).mean()))
    r['M panel store + B2 (2->3)'] = round(float((s[:, mid, 3] - s[:, mid, 2]).mean()))
    r['M tile period'] = round(float((s[:, 2:nt, 0] - s[:, 1:nt - 1, 0]).mean()))
    r['N wait B1+B2 (4->5)'] = round(float((s[:, mid, 5] - s[:, mid, 4]).mean()))
    r['N issue loads (5->9)'] = round(float((s[:, mid, 9] - s[:, mid, 5]).mean()))
    r['N list 0 (9->6)'] = round(float((s[:, mid, 6] - s[:, mid, 9]).mean()))
    r['N list 0 (5->6)'] = round(float((s[:, mid, 6] - s[:, mid, 5]).mean()))
    r['N list 1 (6->7)'] = round(float((s[:, mid, 7] - s[:, mid, 6]).mean()))
    r['N tile period'] = round(float((s[:, 2:nt, 4] - s[:, 1:nt - 1, 4]).mean()))
    r['first M stamp -> last N stamp'] = round(float((s[:, nt - 1, 7] - s[:, 0, 0]).mean()))
    out[f'cbp_bin={v}'] = r
lib.hk_lab_set_cbf_stamps(None)
lib.hk_tuning_set(b'cbp_bin', -1)
print(json.dumps(out, indent=1))
od = os.path.join(ROOT, 'gpurun_out')
if os.path.isdir(od):
    open(os.path.join(od, f'cbf_lab_B{B}.json'), 'w').write(json.dumps(out, indent=1))
