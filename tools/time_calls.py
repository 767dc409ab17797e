#!/usr/bin/env python3
"""Tuning aid: time single MSDeformAttn calls (HIP events around repeated launches) for chosen variants / options.

    python tools/time_calls.py [--calls E,Dd] [--loc init,sigma4,uniform] [--fwd 0,1,2] [--bwd 0,1,2,3] [--opt k=v ...]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from richsem_amd import _lib, workload as W   # noqa: E402
from richsem_amd import MultiScaleDeformableAttention as MSDA   # noqa: E402


def time_fn(fn, reps):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", default="E,Dd")
    ap.add_argument("--loc", default="init,sigma4,uniform")
    ap.add_argument("--fwd", default="")
    ap.add_argument("--bwd", default="1,4")
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--sets", type=int, default=4, help="distinct tensor sets cycled through (cache residency)")
    args = ap.parse_args()
    _lib.load()
    _lib.set_option("locality_monitor", 0)
    for kv in args.opt:
        k, v = kv.split("=")
        _lib.set_option(k, int(v))
    mk = {"E": W.call_E, "Dd": W.call_Dd, "Em": W.call_Em}
    for cname in args.calls.split(","):
        call = mk[cname](2)
        base = [W.make_inputs(call, "init", seed=s, device="cuda") for s in range(args.sets)]
        for mode in args.loc.split(","):
            locs = [W.make_loc(call, mode, seed=s, device="cuda") for s in range(args.sets)]
            state = {"i": 0}

            def run(kind):
                i = state["i"] = (state["i"] + 1) % args.sets
                t = base[i]
                if kind == "fwd":
                    MSDA.ms_deform_attn_forward(t["value"], t["shapes"], t["lsi"], locs[i], t["aw"], 64)
                else:
                    MSDA.ms_deform_attn_backward(t["value"], t["shapes"], t["lsi"], locs[i], t["aw"], t["grad_out"], 64)
            for fv in [int(x) for x in args.fwd.split(",") if x]:
                _lib.set_option("fwd_variant", fv)
                us = time_fn(lambda: run("fwd"), args.reps)
                print(f"{cname:3s} {mode:8s} fwd variant {fv}: {us:8.1f} us  ({call.bytes_fwd() / (us * 1e-6) / 8e12:.3f} of 8 TB/s)", flush=True)
            for bv in [int(x) for x in args.bwd.split(",") if x]:
                _lib.set_option("bwd_variant", bv)
                us = time_fn(lambda: run("bwd"), args.reps)
                print(f"{cname:3s} {mode:8s} bwd variant {bv}: {us:8.1f} us  ({call.bytes_bwd() / (us * 1e-6) / 8e12:.3f} of 8 TB/s)", flush=True)


if __name__ == "__main__":
    main()
