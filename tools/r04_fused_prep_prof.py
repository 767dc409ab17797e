import sys, torch
from richsem_amd import _lib, workload as W
from richsem_amd.functions import MSDeformAttnFusedFunction
call = W.call_Dd(2)
shapes, lsi = W.level_tensors(call, "cuda")
N, Lq, S, M, D, L, P = call.N, call.Lq, call.S, call.M, call.D, call.L, call.P
for dt in (torch.float32, torch.bfloat16):
    value = torch.randn(N, S, M, D, device="cuda").to(dt)
    qproj = torch.randn(N, Lq, M * L * P * 3, device="cuda").to(dt)
    ref = (torch.rand(N, Lq, L, 4, device="cuda") * 0.5 + 0.2)
    for fused in (0, 1):
        _lib.set_option("fwd_prep_fused", fused)
        with torch.no_grad():
            for _ in range(30):
                MSDeformAttnFusedFunction.apply(value, shapes, lsi, qproj, ref, M, L, P, 64)
        torch.cuda.synchronize()
