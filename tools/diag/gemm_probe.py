import sys, torch
sys.path.insert(0, '/root/repo')
from diga_amd.model.networks.MixTransfomer import _Ops
ops = _Ops(torch.device('cuda'))
def t(m,n,k,f32=False,reps=10):
    x = torch.randn((m,k),device='cuda').half(); w = torch.randn((n,k),device='cuda').half()
    out = torch.empty((m,n),device='cuda',dtype=torch.float32 if f32 else torch.float16)
    ops.gemm(x,w,None,n,out_f32=f32,out=out); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.gemm(x,w,None,n,out_f32=f32,out=out)
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)/reps*1e3
    print(f"{m}x{n}x{k} f32={f32}: {us:.1f} us {2.0*m*n*k/us/1e6:.0f} TFLOP/s")
for k in (320, 640, 1280, 2560, 5120):
    t(36864,1280,k)
for n in (128, 256, 512, 1280, 2560):
    t(36864,n,320)
t(8192,8192,8192)
