#!/usr/bin/env python3
"""Tuning aid: per-stage shader-clock times of the row-band backward kernel's workgroups (msda_debug_stamps), by level."""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from richsem_amd import _lib, workload as W   # noqa: E402
from richsem_amd import MultiScaleDeformableAttention as MSDA   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--loc", default="init")
ap.add_argument("--opt", action="append", default=[])
args = ap.parse_args()
lib = _lib.load()
_lib.set_option("bwd_variant", 5)
for kv in args.opt:
    k, v = kv.split("=")
    _lib.set_option(k, int(v))
call = W.call_Dd(2)
t = W.make_inputs(call, "init", seed=0, device="cuda")
loc = W.make_loc(call, args.loc, seed=0, device="cuda")
run = lambda: MSDA.ms_deform_attn_backward(t["value"], t["shapes"], t["lsi"], loc, t["aw"], t["grad_out"], 64)
for _ in range(3):
    run()
buf = torch.zeros(4096 * 16, dtype=torch.int64, device="cuda")
lib.msda_debug_stamps(ctypes.c_void_p(buf.data_ptr()))
run()
torch.cuda.synchronize()
lib.msda_debug_stamps(None)
s = buf.view(-1, 16).cpu()
s = s[s[:, 0] > 0].double()
t0 = s[:, 0].min()
print(f"{len(s)} workgroups; kernel span {(s[:, 4].max() - t0) / 100:.1f} us (100 MHz clock)")
ent = s[:, 5].long()
for name, lo, hi in (("all", 0, 10 ** 6),):
    pass
zero, scan, red, flush = s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2], s[:, 4] - s[:, 3]
tot = s[:, 4] - s[:, 0]
print(f"per workgroup (us): total {tot.mean() / 100:.2f} (max {tot.max() / 100:.2f})  zero {zero.mean() / 100:.2f}  scan {scan.mean() / 100:.2f}  "
      f"reduce {red.mean() / 100:.2f} (max {red.max() / 100:.2f})  flush {flush.mean() / 100:.2f}; listed items {s[:, 6].mean():.0f} (max {s[:, 6].max():.0f})")
# by entry index (levels are laid out in order: the first entries are level 0)
import collections
by = collections.defaultdict(list)
for i in range(len(s)):
    by[int(ent[i])].append(i)
keys = sorted(by)
for lo in range(0, len(keys), max(1, len(keys) // 12)):
    idx = [i for k in keys[lo:lo + max(1, len(keys) // 12)] for i in by[k]]
    idx = torch.tensor(idx)
    print(f"entries {keys[lo]:3d}..: total {tot[idx].mean() / 100:6.2f} zero {zero[idx].mean() / 100:5.2f} scan {scan[idx].mean() / 100:5.2f} "
          f"reduce {red[idx].mean() / 100:6.2f} flush {flush[idx].mean() / 100:5.2f} items {s[idx, 6].mean():6.0f} start {(s[idx, 0].mean() - t0) / 100:6.1f}")
