#!/usr/bin/env python3
"""Diagnostic: where a workgroup of the LDS-window kernels spends its cycles (s_memtime stamps per stage).
    python tools/stage_stamps.py --call E --kernel fwd|bwd
Stamps: 0 start, 1 header built, then kernel-specific stage ends (see msda_tiled.h)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from richsem_amd import _lib, workload as W                       # noqa: E402
from richsem_amd import MultiScaleDeformableAttention as MSDA    # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--call", default="E")
ap.add_argument("--kernel", default="fwd")
ap.add_argument("--loc", default="init")
ap.add_argument("--which", type=int, default=0, help="1 = scatter only, 2 = gather only")
ap.add_argument("--set", action="append", default=[], help="option=value")
args = ap.parse_args()
call = {"E": W.call_E, "Em": W.call_Em}[args.call](2)
t = W.make_inputs(call, args.loc, seed=0, device="cuda")
lib = _lib.load()
for kv in args.set:
    k, v = kv.split('=')
    _lib.set_option(k, int(v))
nwg = 20000
buf = torch.zeros(nwg * 16, dtype=torch.int64, device="cuda")


def run():
    if args.kernel == "fwd":
        MSDA.ms_deform_attn_forward(t["value"], t["shapes"], t["lsi"], t["loc"], t["aw"], 64)
    else:
        MSDA.ms_deform_attn_backward(t["value"], t["shapes"], t["lsi"], t["loc"], t["aw"], t["grad_out"], 64)


run()
torch.cuda.synchronize()
_lib.set_option('tile_debug', args.which << 4)
lib.msda_debug_stamps(buf.data_ptr())
run()   # for bwd both kernels write the same rows: the later one (gather) wins where grids overlap
torch.cuda.synchronize()
lib.msda_debug_stamps(None)
_lib.set_option('tile_debug', 0)
s = buf.view(nwg, 16).cpu()
used = s[:, 0] != 0
s = s[used].double()
print(f"{int(used.sum())} workgroups stamped")
prev = s[:, 0]
for i in range(1, 16):
    cur = s[:, i]
    ok = cur != 0
    if not ok.any():
        break
    d = (cur - prev)[ok]
    print(f"stage {i - 1}->{i}: mean {d.mean():9.0f} ticks  median {d.median():9.0f}  (n={int(ok.sum())})")
    prev = cur
tot = (prev - s[:, 0])
print(f"total per workgroup (or per last item of a persistent workgroup): mean {tot.mean():.0f} shader cycles")
