#!/usr/bin/env python3
"""Tuning aid: how the route pass's workgroups differ (msda_debug_stamps rows 1024..): totals by rank, by XCD, by number of work items."""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from richsem_amd import _lib, workload as W   # noqa: E402
from richsem_amd import MultiScaleDeformableAttention as MSDA   # noqa: E402


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
    r = buf.view(-1, 16)[1024:2048].cpu().double()
    n = int((r[:, :8].sum(1) > 0).sum())
    r = r[:n]
    tot = r[:, :8].sum(1)
    names = ["prologue", "operands+barrier", "A ranks", "barrier", "B scan", "C stores", "D announce", "barrier"]
    print(f"{n} workgroups; total cycles: mean {tot.mean():.0f}  min {tot.min():.0f}  p10 {tot.quantile(0.1):.0f}  p50 {tot.quantile(0.5):.0f}  p90 {tot.quantile(0.9):.0f}  max {tot.max():.0f}")
    order = tot.argsort()
    for label, idx in (("fastest 32", order[:32]), ("slowest 32", order[-32:])):
        print(f"  {label}: blockIdx sample {sorted(idx.tolist())[:12]} ...")
        print("     " + "  ".join(f"{nm} {r[idx, i].mean():.0f}" for i, nm in enumerate(names)))
    for x in range(8):
        sel = torch.arange(n) % 8 == x
        print(f"  XCD {x}: mean {tot[sel].mean():.0f}  max {tot[sel].max():.0f}")
    for lo in range(0, n, 64):
        print(f"  blocks {lo:4d}..{lo + 63:4d}: mean {tot[lo:lo + 64].mean():.0f}")


if __name__ == "__main__":
    main()
