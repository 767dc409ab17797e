#!/usr/bin/env python3
"""Tuning aid: bf16-storage backward of call E per sampling distribution and backward variant."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from richsem_amd import _lib, workload as W                       # noqa: E402
from richsem_amd import MultiScaleDeformableAttention as MSDA    # noqa: E402

_lib.load()
_lib.set_option("locality_monitor", 0)
call = W.call_E(2)
for mode in ("init", "sigma4", "uniform"):
    t = W.make_inputs(call, mode, seed=0, device="cuda")
    v, go = t["value"].to(torch.bfloat16), t["grad_out"].to(torch.bfloat16)
    for variant in (4, 1):
        _lib.set_option("bwd_variant", variant)
        fn = lambda: MSDA.ms_deform_attn_backward(v, t["shapes"], t["lsi"], t["loc"], t["aw"], go, 64)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            fn()
        b.record()
        torch.cuda.synchronize()
        print(f"E bf16 {mode:8s} bwd variant {variant}: {a.elapsed_time(b) / 10 * 1e3:8.1f} us")
