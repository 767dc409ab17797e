#!/usr/bin/env python3
"""Tuning aid: forward + backward time of the MSDeformAttn MODULE (four projections + the operator) at BASELINE's encoder and
decoder call shapes, fused module path (one 256->384 GEMM, library softmax / location / mask kernels) vs the reference's
op-by-op sequence."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from richsem_amd import _lib, workload as W   # noqa: E402
from richsem_amd.modules import MSDeformAttn   # noqa: E402


def main():
    _lib.load()
    torch.manual_seed(0)
    for name, call in (("E", W.call_E(2)), ("Dd", W.call_Dd(2))):
        shapes, lsi = W.level_tensors(call, "cuda")
        mod = MSDeformAttn(256, 4, 8, 4).cuda()
        query = torch.randn(call.N, call.Lq, 256, device="cuda", requires_grad=True)
        src = torch.randn(call.N, call.S, 256, device="cuda", requires_grad=True)
        if call.encoder:
            ref = W.encoder_reference_points(call).cuda()[None, :, None, :].expand(call.N, call.Lq, 4, 2).contiguous()
        else:
            ref = torch.rand(call.N, call.Lq, 4, 4, device="cuda") * 0.5 + 0.2
        mask = torch.zeros(call.N, call.S, dtype=torch.bool, device="cuda")
        mask.view(-1)[::122] = True
        gout = torch.randn(call.N, call.Lq, 256, device="cuda")
        for fused, dt in ((False, torch.float32), (True, torch.float32), (True, torch.bfloat16)):
            mod.fused = fused
            q, s_, go = (query.detach().to(dt).requires_grad_(True), src.detach().to(dt).requires_grad_(True), gout.to(dt))

            def step():
                out = mod(q, ref, s_, shapes, lsi, mask)
                out.backward(go)
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                step()
            b.record()
            torch.cuda.synchronize()
            label = ("fused   " if fused else "op-by-op") + (" bf16" if dt == torch.bfloat16 else " fp32")
            print(f"{name}: module forward+backward, {label}: {a.elapsed_time(b) / 20 * 1e3:8.1f} us", flush=True)

if __name__ == "__main__":
    main()
