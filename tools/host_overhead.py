#!/usr/bin/env python3
"""Tuning aid: host time per call (Python shim + ctypes + the library's planning) against the GPU time of the call."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from richsem_amd import _lib, workload as W                       # noqa: E402
from richsem_amd import MultiScaleDeformableAttention as MSDA    # noqa: E402

_lib.load()
for name, mk in (("Dd", W.call_Dd), ("E", W.call_E)):
    t = W.make_inputs(mk(2), "init", seed=0, device="cuda")
    for kind in ("fwd", "bwd"):
        if kind == "fwd":
            fn = lambda: MSDA.ms_deform_attn_forward(t["value"], t["shapes"], t["lsi"], t["loc"], t["aw"], 64)
        else:
            fn = lambda: MSDA.ms_deform_attn_backward(t["value"], t["shapes"], t["lsi"], t["loc"], t["aw"], t["grad_out"], 64)
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        t_issue = (time.perf_counter() - t0) / n
        torch.cuda.synchronize()
        t_total = (time.perf_counter() - t0) / n
        print(f"{name} {kind}: host issue {t_issue * 1e6:7.1f} us per call, wall incl. GPU {t_total * 1e6:7.1f} us per call")
