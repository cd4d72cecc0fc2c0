import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from diga_amd.model import seg_model_noaux as sm
from diga_amd.model import norm as dn
torch.manual_seed(0)
se = sm.SEBlock(64, 16).cuda()
x0 = torch.randn(2, 64, 9, 7, device="cuda").contiguous(memory_format=torch.channels_last)
def run(x):
    for p in se.parameters(): p.grad = None
    x.grad = None
    y = se(x)
    (y * y).sum().backward()
xe = x0.clone().requires_grad_()
run(xe)
ref = [p.grad.clone() for p in se.parameters()] + [xe.grad.clone()]
xs = x0.clone().requires_grad_()
run(xs)  # warm
torch.cuda.synchronize()
for p in se.parameters(): p.grad = None
xs.grad = None
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    y = se(xs)
    (y * y).sum().backward()
g.replay(); torch.cuda.synchronize()
got = [p.grad.clone() for p in se.parameters()] + [xs.grad.clone()]
for n, a, b in zip([n for n, _ in se.named_parameters()] + ["x"], ref, got):
    print(n, float((a - b).abs().max()), float(a.abs().max()), float(b.abs().max()))
# variant: pure torch linear path only
lin = torch.nn.Sequential(torch.nn.Linear(64, 4), torch.nn.ReLU(inplace=True), torch.nn.Linear(4, 64), torch.nn.Sigmoid()).cuda()
z0 = torch.randn(2, 64, device="cuda")
def run2(z):
    for p in lin.parameters(): p.grad = None
    lin(z).pow(2).sum().backward()
ze = z0.clone().requires_grad_(); run2(ze); ref2 = [p.grad.clone() for p in lin.parameters()]
zs = z0.clone().requires_grad_(); run2(zs); torch.cuda.synchronize()
for p in lin.parameters(): p.grad = None
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    lin(zs).pow(2).sum().backward()
g2.replay(); torch.cuda.synchronize()
for (n, _), a, p in zip(lin.named_parameters(), ref2, lin.parameters()):
    print("torch-only", n, float((a - p.grad).abs().max()), float(a.abs().max()))
