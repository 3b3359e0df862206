"""Where the ROI-refinement backward spends its time: cycle stamps (s_memtime) of thread 0 of the first 64 workgroups
of image 0, from the instrumented build (make -C hawkeye_amd/csrc lab).  B = 16, C = 512, 56 x 56 (the AP-CNN shape).
    python tools/roi_lab.py"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

lib = ctypes.CDLL(os.path.join(ROOT, 'hawkeye_amd', 'csrc', 'libhawkeye_hip_lab.so'))
P, I = ctypes.c_void_p, ctypes.c_int
lib.hk_roi_crop_resize_bwd.argtypes = [P, P, P, P, I, I, I, I, I, P]
lib.hk_lab_set_roi_stamps.argtypes = [P]
lib.hk_tuning_set.argtypes = [ctypes.c_char_p, I]

dev = torch.device('cuda:0')
B, C, H, W = 16, 512, 56, 56
dy = torch.randn(B, C, H, W, device=dev)
dx = torch.empty_like(dy)
g = torch.Generator().manual_seed(0)
x1 = torch.randint(0, 20, (B,), generator=g).float()
y1 = torch.randint(0, 20, (B,), generator=g).float()
box = torch.stack([x1, y1, x1 + torch.randint(24, 36, (B,), generator=g), y1 + torch.randint(24, 36, (B,), generator=g)], 1).to(dev)
drop = torch.tensor([[0., 0., -1., -1.]] * B, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def p(t):
    return ctypes.c_void_p(t.data_ptr())


def run():
    assert lib.hk_roi_crop_resize_bwd(p(dy), p(box), p(drop), p(dx), B, C, H, W, 0, st) == 0


lib.hk_lab_set_roi_stamps(None)
for _ in range(3):
    run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
stamps = torch.zeros(64, 32, 8, dtype=torch.int64, device=dev)
lib.hk_lab_set_roi_stamps(p(stamps))
run()
torch.cuda.synchronize()
lib.hk_lab_set_roi_stamps(None)
s = stamps.cpu().double()
pro = s[:, 31, :]
out = {'us_uninstrumented': round(us, 1), 'box0': box[0].tolist(),
       'prologue_cycles': {'tables + window size': round(float((pro[:, 1] - pro[:, 0]).mean())),
                           'pixel geometry + weights': round(float((pro[:, 2] - pro[:, 1]).mean())),
                           'whole workgroup': round(float((pro[:, 3] - pro[:, 0]).mean()))},
       'per_map_cycles': {}}
names = ['barrier + LDS store of the staged map + barrier', 'issue next loads + gather (crop pixels) -> LDS image',
         'barrier + 16-byte stores of the LDS image']
nm = 6
for i, n in enumerate(names):
    out['per_map_cycles'][n] = round(float((s[:, 1:nm, i + 1] - s[:, 1:nm, i]).mean()))
out['per_map_cycles']['whole map'] = round(float((s[:, 2:nm, 0] - s[:, 1:nm - 1, 0]).mean()))
print(json.dumps(out, indent=1))
