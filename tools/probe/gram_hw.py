import ctypes, torch, os, sys
c = ctypes
path = sys.argv[1]
lib = c.CDLL(path)
lib.hk_bcnn_gram_norm.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_int, c.c_int, c.c_int, c.c_void_p]
lib.hk_bcnn_gram_norm.restype = c.c_int
dev = torch.device('cuda:0')
st = torch.cuda.current_stream().cuda_stream
for HW in (196, 144, 100, 64):
    B, C = 64, 512
    x = torch.relu(torch.randn(B, C, HW, device=dev)); y = torch.empty(B, C * C, device=dev); inv = torch.rand(B, device=dev) + 0.5
    fn = lambda: lib.hk_bcnn_gram_norm(x.data_ptr(), inv.data_ptr(), y.data_ptr(), B, C, HW, st)
    ts = []
    for r in range(4):
        for _ in range(3): assert fn() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 40 * 1e3)
    us = sorted(ts)[1]
    fl = 2.0 * B * C * C * HW * 36 / 64
    print(f'{os.path.basename(path)} HW={HW}: {us:.1f} us, executed {fl/us/1e6:.1f} TF/s = {fl/us/1e6/157.3:.3f}')
