"""Where the CBP row-scatter binning kernel spends its time: cycle stamps (s_memtime) of thread 0 of the workgroups of the
first eight images, from the instrumented build (make -C hawkeye_amd/csrc lab).  B = 64, C = 512, D = 6000, 14 x 14.
    python tools/cbp_lab.py"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import hawkeye_amd.functional as F

lib = ctypes.CDLL(os.path.join(ROOT, 'hawkeye_amd', 'csrc', 'libhawkeye_hip_lab.so'))
P, I = ctypes.c_void_p, ctypes.c_int
lib.hk_cbp_ws_bytes.restype = ctypes.c_size_t
lib.hk_cbp_ws_bytes.argtypes = [I, I, I, I]
lib.hk_cbp_fwd.argtypes = [P, P, P, P, P, I, I, I, I, P, ctypes.c_size_t, P]
lib.hk_lab_set_cbp_stamps.argtypes = [P]
dev = torch.device('cuda:0')
B, C, HW, D = 64, 512, 196, 6000
plan = F.CbpPlan(*F.sketch_hashes(C, C, D), D, dev)            # (the plan blob is built by the product library: same layout)
x = torch.relu(torch.randn(B, C, HW, device=dev))
y, craw, inv = torch.empty(B, D, device=dev), torch.empty(B, D, device=dev), torch.empty(B, device=dev)
nws = lib.hk_cbp_ws_bytes(B, C, HW, D)
ws = torch.empty(nws, dtype=torch.uint8, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr())
run = lambda: lib.hk_cbp_fwd(p(x), p(plan.blob), p(y), p(craw), p(inv), B, C, HW, D, p(ws), nws, st)
lib.hk_lab_set_cbp_stamps(None)
for _ in range(3):
    assert run() == 0
stamps = torch.zeros(64, 32, 8, dtype=torch.int64, device=dev)
lib.hk_lab_set_cbp_stamps(p(stamps))
assert run() == 0
torch.cuda.synchronize()
lib.hk_lab_set_cbp_stamps(None)
s = stamps.cpu().double()[:, 1:15, :]                             # steady-state blocks
names = ['issue next block loads', 'row 0: sketch + bins + barrier', 'row 1', 'row 2', 'row 3', 'LDS store of the next block + barrier']
out = {n: round(float((s[:, :, i + 1] - s[:, :, i]).mean())) for i, n in enumerate(names)}
out['whole block of 4 rows'] = round(float((s[:, 1:, 0] - s[:, :-1, 0]).mean()))
full = stamps.cpu().double()
out['workgroup: first to last stamp'] = round(float((full[:, 15, 6] - full[:, 0, 0]).mean()))
print(json.dumps(out, indent=1))
