#!/usr/bin/env python3
"""Round 4, SURVEY section 8f rank 1 ("head-major value layout"): the E forward with value stored (N, M, S, D) -- a head's pixel rows contiguous --
against the reference's (N, S, M, D), both forward kernels (LDS-window and direct), same inputs, same results.  tile_debug bit 7 (128) makes the
forward kernels read value as head-major (a measured experiment: nothing in the product writes that layout)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from richsem_amd import _lib, workload as W   # noqa: E402
from richsem_amd import MultiScaleDeformableAttention as MSDA   # noqa: E402

_lib.load()
_lib.set_option("locality_monitor", 0)
call = W.call_E(2)
sets = [W.make_inputs(call, "init", seed=s, device="cuda") for s in range(6)]
for mode in ("init", "uniform"):
    locs = [W.make_loc(call, mode, seed=s, device="cuda") for s in range(6)]
    for variant, name in ((2, "LDS-window kernel"), (1, "direct kernel")):
        _lib.set_option("fwd_variant", variant)
        res = {}
        for hm in (0, 1, 0, 1):
            _lib.set_option("tile_debug", 128 * hm)
            vals = [t["value"].permute(0, 2, 1, 3).contiguous().view_as(t["value"]) if hm else t["value"] for t in sets]      # (the memory of (N, M, S, D) under the shim's shape check)
            outs = None
            for _ in range(3):
                for i, t in enumerate(sets):
                    o = MSDA.ms_deform_attn_forward(vals[i], t["shapes"], t["lsi"], locs[i], t["aw"], 64)
                    if i == 0:
                        outs = o
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                for i, t in enumerate(sets):
                    MSDA.ms_deform_attn_forward(vals[i], t["shapes"], t["lsi"], locs[i], t["aw"], 64)
            b.record()
            torch.cuda.synchronize()
            us = a.elapsed_time(b) / 30 * 1e3
            res.setdefault(hm, []).append(us)
            if hm == 0:
                ref = outs.clone()
            else:
                err = float((outs - ref).abs().max()) / float(ref.abs().max())
                assert err < 1e-6, err
        _lib.set_option("tile_debug", 0)
        print(f"E forward, loc-{mode}, {name}: (N,S,M,D) {min(res[0]):.1f} us   head-major (N,M,S,D) {min(res[1]):.1f} us   (6 tensor sets, 30 calls; same output)", flush=True)
