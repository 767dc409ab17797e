import os, sys, torch
sys.path.insert(0, os.getcwd())
from richsem_amd.functions.ffn import FusedFFNFunction
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/reps*1e3
T,D,Fh=44646,256,2048
x=torch.randn(T,D,device="cuda").bfloat16().requires_grad_(True)
w1=(torch.randn(Fh,D,device="cuda")*D**-0.5).bfloat16().requires_grad_(True); w2=(torch.randn(D,Fh,device="cuda")*Fh**-0.5).bfloat16().requires_grad_(True)
b1=torch.zeros(Fh,device="cuda",requires_grad=True); b2=torch.zeros(D,device="cuda",requires_grad=True)
gw=torch.ones(D,device="cuda",requires_grad=True); gb=torch.zeros(D,device="cuda",requires_grad=True)
g=torch.randn(T,D,device="cuda").bfloat16()
def fwd(): return FusedFFNFunction.apply(x,w1,b1,w2,b2,gw,gb,1e-5)
def both():
    for p in (x,w1,w2,b1,b2,gw,gb): p.grad=None
    fwd().backward(g)
tf=timeit(fwd); tb=timeit(both)
print(f"fused FFN forward {tf:.0f} us, forward+backward {tb:.0f} us -> backward {tb-tf:.0f} us")
h=None
def gemms():
    x2=x.detach(); h=torch.relu(torch.addmm(b1.detach().bfloat16(), x2, w1.detach().t())); gy=g
    a=gy.t()@h; gh=(gy@w2.detach())*(h>0); c=gh.t()@x2; d=gh@w1.detach(); return a,c,d
print(f"the backward's five bf16 GEMMs + relu mask alone: {timeit(gemms):.0f} us")
