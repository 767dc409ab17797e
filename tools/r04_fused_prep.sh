#!/bin/bash
# round 4, f1: the fused prep + gather forward against the two-kernel form, decoder-shaped module forward + backward
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_gpu_module.py tests/test_gpu_layers.py tests/test_gpu_step.py tests/test_gpu_ddp.py -x -q -m gpu 2>&1 | tail -4
python - <<'PY'
import torch, time
from richsem_amd import _lib, workload as W
from richsem_amd.functions import MSDeformAttnFusedFunction
call = W.call_Dd(2)
shapes, lsi = W.level_tensors(call, "cuda")
N, Lq, S, M, D, L, P = call.N, call.Lq, call.S, call.M, call.D, call.L, call.P
for dt in (torch.float32, torch.bfloat16):
    work = torch.float32
    value = torch.randn(N, S, M, D, device="cuda").to(dt)
    qproj = torch.randn(N, Lq, M * L * P * 3, device="cuda").to(dt)
    ref = (torch.rand(N, Lq, L, 4, device="cuda") * 0.5 + 0.2).to(work)
    for fused in (0, 1, 0, 1):
        _lib.set_option("fwd_prep_fused", fused)
        with torch.no_grad():
            for _ in range(5):
                MSDeformAttnFusedFunction.apply(value, shapes, lsi, qproj, ref, M, L, P, 64)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(200):
                MSDeformAttnFusedFunction.apply(value, shapes, lsi, qproj, ref, M, L, P, 64)
            b.record()
            torch.cuda.synchronize()
        print(f"Dd forward (prep + gather), {dt}, fwd_prep_fused={fused}: {a.elapsed_time(b) / 200 * 1e3:.1f} us per call (wall, 200 calls back to back)")
PY
