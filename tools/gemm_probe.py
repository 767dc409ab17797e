import torch
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/reps*1e3
T=44646
x=torch.randn(T,256,device="cuda"); 
for N in (256,384,2048):
    w=torch.randn(N,256,device="cuda"); b=torch.randn(N,device="cuda")
    t=timeit(lambda: torch.nn.functional.linear(x,w,b)); fl=2*T*256*N
    x16,w16,b16=x.bfloat16(),w.bfloat16(),b.bfloat16()
    t16=timeit(lambda: torch.nn.functional.linear(x16,w16,b16))
    g=torch.randn(T,N,device="cuda")
    tdx=timeit(lambda: g@w); tdw=timeit(lambda: g.t()@x)
    print(f"N={N}: fp32 linear {t:.1f} us ({fl/t/1e6:.0f} TFLOP/s)  bf16 {t16:.1f} us ({fl/t16/1e6:.0f});  fp32 dX {tdx:.1f} us  dW {tdw:.1f} us")
