"""Diagnostic: where does the first BCNN step at the metric's shape spend its time?"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd.miopen_cache import use_in_tree_cache
print('miopen dir', use_in_tree_cache(), flush=True)
import torch
T0 = time.time()
def log(*a):
    print(f'[{time.time()-T0:7.1f}s]', *a, flush=True)
import hawkeye_amd.functional as F
from bench import build_model
dev = torch.device('cuda:0')
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
size = int(sys.argv[2]) if len(sys.argv) > 2 else 448
log('start', bs, size)
x = torch.relu(torch.randn(bs, 512, 14, 14, device=dev)).requires_grad_(True)
y = F.bilinear_pool(x); torch.cuda.synchronize(); log('pool fwd ok')
y.backward(torch.randn_like(y)); torch.cuda.synchronize(); log('pool bwd ok')
m = build_model('BCNN', 200).to(dev); log('model on device')
img = torch.randn(bs, 3, size, size, device=dev)
with torch.no_grad():
    f = m.backbone(img); torch.cuda.synchronize(); log('backbone fwd (no grad)', tuple(f.shape))
    f = m.backbone(img); torch.cuda.synchronize(); log('backbone fwd 2nd')
out = m(img); torch.cuda.synchronize(); log('model fwd')
loss = out.sum(); loss.backward(); torch.cuda.synchronize(); log('model bwd')
out = m(img); out.sum().backward(); torch.cuda.synchronize(); log('model fwd+bwd 2nd')
