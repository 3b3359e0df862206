import os, sys
sys.path.insert(0, os.getcwd())
import copy, torch
from hawkeye_amd.model.backbone import vgg16
def rel(a, b): return float((a.double() - b.double()).norm() / b.double().norm())
torch.manual_seed(3)
fused = vgg16(pretrained=False).features.cuda().to(memory_format=torch.channels_last)
plain = torch.nn.Sequential(*copy.deepcopy(fused).children())
x = torch.randn(4, 3, 96, 64, device='cuda').contiguous(memory_format=torch.channels_last)
wt = torch.randn(4, 512, 3, 2, device='cuda')
def run(net):
    xi = x.clone().requires_grad_(True)
    for p in net.parameters(): p.grad = None
    y = net(xi); (y * wt).sum().backward()
    torch.cuda.synchronize()
    return xi.grad.clone(), [p.grad.clone() for p in net.parameters()]
# float64 reference on the CPU
ref = copy.deepcopy(plain).cpu().double()
xr = x.cpu().double().contiguous().requires_grad_(True)
(ref(xr) * wt.cpu().double()).sum().backward()
g64 = xr.grad
for i in range(8):
    gf, pf = run(fused); gp, pp = run(plain)
    print(i, 'fused vs f64 %.2e  plain vs f64 %.2e  fused vs plain %.2e | first conv dW fused %.2e plain %.2e' % (
        rel(gf.cpu(), g64), rel(gp.cpu(), g64), rel(gf, gp), rel(pf[0].cpu(), list(ref.parameters())[0].grad), rel(pp[0].cpu(), list(ref.parameters())[0].grad)))
