"""What the fp32 matrix pipe sustains with nothing else going on (tools/probe/mfma_peak.hip): TFLOP/s of back-to-back
v_mfma_f32_16x16x4_f32 / 32x32x2 on random operands, 8 and 16 waves per CU, for launches of ~0.1 ms and ~2 ms, cold and
after 20 ms of sustained load.  The datasheet figure (157.3 TF/s) assumes 2.4 GHz; this is the number a kernel's `frac`
can actually reach.      python tools/mfma_peak.py > gpurun_out/mfma_peak.json"""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, 'tools', 'probe', 'libhk_probe.so'))
lib.hk_probe_mfma.restype = ctypes.c_longlong
lib.hk_probe_mfma.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device('cuda:0')
rows = []
for zero in (False, True):
    src = torch.zeros(65536, device=dev) if zero else torch.randn(65536, device=dev)
    out = torch.empty(512 * 512, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for kind, name, fl in ((0, '16x16x4 x8 acc', 2048.0), (1, '16x16x4 x16 acc', 2048.0), (2, '32x32x2 x4 acc', 4096.0)):
        for blocks, threads in ((256, 512), (512, 512), (256, 256)):
            for iters in (400, 8000):
                def call():
                    n = lib.hk_probe_mfma(src.data_ptr(), out.data_ptr(), kind, blocks, threads, iters, st)
                    assert n > 0
                    return n
                nm = call()
                torch.cuda.synchronize()
                reps = 20 if iters < 1000 else 6
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                ev[0].record()
                for _ in range(reps):
                    call()
                ev[1].record()
                for _ in range(reps):
                    call()
                ev[2].record()
                torch.cuda.synchronize()
                waves = blocks * threads // 64
                for tag, ms in (('first', ev[0].elapsed_time(ev[1]) / reps), ('second', ev[1].elapsed_time(ev[2]) / reps)):
                    tf = nm * fl * waves / (ms * 1e-3) / 1e12
                    rows.append({'operands': 'zeros' if zero else 'randn', 'mfma': name, 'blocks': blocks, 'threads': threads,
                                 'iters': iters, 'round': tag, 'us': round(ms * 1e3, 1), 'tflops': round(tf, 1),
                                 'frac_of_157.3': round(tf / 157.3, 3)})
json.dump(rows, sys.stdout, indent=0)
