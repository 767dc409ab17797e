#!/usr/bin/env python3
"""A/B of the window forward kernel's fill (window rows by LDS DMA; tile_debug bit 8: load -> register -> LDS store as before), E shape,
fp32, three location patterns: bit-equality of the outputs, then HIP-event times."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from richsem_amd import _lib, workload as W                        # noqa: E402
from richsem_amd import MultiScaleDeformableAttention as MSDA        # noqa: E402


def time_fn(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


_lib.load()
_lib.set_option("locality_monitor", 0)
_lib.set_option("fwd_variant", 2)
call = W.call_E(2)
sets = [W.make_inputs(call, "init", seed=s, device="cuda") for s in range(4)]
for mode in ("init", "sigma4"):
    locs = [W.make_loc(call, mode, seed=s, device="cuda") for s in range(4)]
    outs = {}
    for dbg in (256, 0):
        _lib.set_option("tile_debug", dbg)
        t = sets[0]
        outs[dbg] = MSDA.ms_deform_attn_forward(t["value"], t["shapes"], t["lsi"], locs[0], t["aw"], 64)
    same = torch.equal(outs[0], outs[256])
    res = []
    for dbg in (256, 0, 256, 0):
        _lib.set_option("tile_debug", dbg)
        st = {"i": 0}

        def run():
            i = st["i"] = (st["i"] + 1) % 4
            t = sets[i]
            MSDA.ms_deform_attn_forward(t["value"], t["shapes"], t["lsi"], locs[i], t["aw"], 64)
        res.append(time_fn(run))
    print(f"{mode:8s} outputs equal: {same};  register-staged fill {min(res[0], res[2]):6.1f} us   LDS-DMA fill {min(res[1], res[3]):6.1f} us", flush=True)
_lib.set_option("tile_debug", 0)
