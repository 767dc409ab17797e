#!/usr/bin/env python3
"""Tuning aid: msda_forward_prep_* on the encoder call E -- prep_forward_kernel + window gather (fwd_prep_fused 0 / 1) against the window
kernel that reads the raw projection itself (2): µs per call by events, f32 and bf16, init-like offsets."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from richsem_amd import _lib, workload as W                       # noqa: E402
from richsem_amd.functions import MSDeformAttnFusedFunction      # noqa: E402
from richsem_amd.modules import get_reference_points              # noqa: E402

call = W.call_E(2)
shapes, lsi = W.level_tensors(call, "cuda")
N, S, M, D, L, P = call.N, call.S, call.M, call.D, call.L, call.P
g = torch.Generator(device="cuda").manual_seed(0)
vr = torch.ones(N, L, 2, device="cuda")
ref = get_reference_points(shapes.tolist(), vr, "cuda").float().contiguous()
_lib.set_option("fwd_variant", 2)
for dt in (torch.float32, torch.bfloat16):
    sets = []
    for i in range(4):
        value = torch.randn(N, S, M, D, device="cuda", generator=g).to(dt)
        qproj = torch.randn(N, S, M * L * P * 3, device="cuda", generator=g)
        qproj[..., :M * L * P * 2] *= 1.5
        sets.append((value, qproj.to(dt)))
    for rnd in range(2):
        for fused in (1, 2):
            _lib.set_option("fwd_prep_fused", fused)
            with torch.no_grad():
                for v, q in sets:
                    MSDeformAttnFusedFunction.apply(v, shapes, lsi, q, ref, M, L, P, 64)
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(10):
                    for v, q in sets:
                        MSDeformAttnFusedFunction.apply(v, shapes, lsi, q, ref, M, L, P, 64)
                b.record()
                torch.cuda.synchronize()
            print(f"{str(dt):16s} fwd_prep_fused={fused}: {a.elapsed_time(b) / 40 * 1e3:7.1f} us per call", flush=True)
