#!/usr/bin/env python3
"""Tuning aid: per-stage shader-clock shares of the routed backward tile kernel (msda_debug_stamps)."""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from richsem_amd import _lib, workload as W   # noqa: E402
from richsem_amd import MultiScaleDeformableAttention as MSDA   # noqa: E402

NAMES = ["item header + value rows", "entries + locations", "ranks", "scan", "placement", "reduce + dots", "combine", "fold + flush", "queue tail", "-"]

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--call", default="E")
    ap.add_argument("--loc", default="init")
    ap.add_argument("--opt", action="append", default=[])
    args = ap.parse_args()
    lib = _lib.load()
    _lib.set_option("locality_monitor", 0)
    _lib.set_option("bwd_variant", 4)
    for kv in args.opt:
        k, v = kv.split("=")
        _lib.set_option(k, int(v))
    call = {"E": W.call_E, "Dd": W.call_Dd, "Em": W.call_Em}[args.call](2)
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
    s = buf.view(-1, 16)[:256].cpu().double()
    tot = s[:, :10].sum(1)
    r = buf.view(-1, 16)[1024:2048].cpu().double()
    r = r[r[:, :8].sum(1) > 0]
    if len(r):
        rt = r[:, :8].sum(1)
        print(f"route pass: {len(r)} workgroups, per workgroup {rt.mean():.0f} cycles (min {rt.min():.0f}, max {rt.max():.0f})")
        for i, n in enumerate(["prologue (tables, zero-fill)", "operands + barrier", "A ranks", "barrier", "B scan", "C record stores", "D announce", "barrier"]):
            print(f"  {n:34s} {r[:, i].mean():10.0f} cycles  {100 * r[:, i].mean() / rt.mean():5.1f} %")
    print(f"{args.call} loc-{args.loc}: per workgroup {tot.mean():.0f} cycles (min {tot.min():.0f}, max {tot.max():.0f})")
    for i, n in enumerate(NAMES):
        print(f"  {n:34s} {s[:, i].mean():10.0f} cycles  {100 * s[:, i].mean() / tot.mean():5.1f} %")
    ng = s[:, 12].mean()
    if ng > 0:
        print(f"  wave 0 inside the walk: {ng:.0f} groups per workgroup; per group: set-up (unit, value rows, first row request) {s[:, 9].mean() / ng:.0f}, "
              f"point loop {s[:, 10].mean() / ng:.0f}, f64 epilogue {s[:, 11].mean() / ng:.0f} cycles")
        steps = s[:, 13].mean()
        if steps > 0:
            print(f"  wave 0's groups: {steps / ng:.1f} points in the longest unit on average -> {s[:, 10].mean() / steps:.0f} cycles per step of the point loop")


if __name__ == "__main__":
    main()
