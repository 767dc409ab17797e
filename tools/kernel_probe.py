#!/usr/bin/env python3
"""Time single MSDeformAttn calls on the GPU under different library options (tuning aid).

    python tools/kernel_probe.py --call E --loc init --reps 5 --set tile_margin=4 --set tile_debug=1
Prints one line per (kind, variant): average kernel time from the library's event log.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from richsem_amd import _lib, workload as W                       # noqa: E402
from richsem_amd import MultiScaleDeformableAttention as MSDA    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--call", default="E", choices=["E", "Dd", "Em"])
    ap.add_argument("--loc", default="init")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--set", action="append", default=[], help="option=value[,value...] (sweep)")
    ap.add_argument("--only", default="both", choices=["fwd", "bwd", "both"])
    ap.add_argument("--jitter", type=float, default=1.0, help="sigma (pixels of the sampled level) of the 'init' pattern")
    ap.add_argument("--stats", action="store_true", help="print the share of points that miss the forward windows")
    args = ap.parse_args()
    call = {"E": W.call_E, "Dd": W.call_Dd, "Em": W.call_Em}[args.call](2)
    t = W.make_inputs(call, args.loc, seed=0, device="cuda", jitter_px=args.jitter)
    if args.stats:
        cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
        _lib.load().msda_debug_stats(cnt.data_ptr())
        _lib.set_option("fwd_variant", 2)    # (in automatic mode the locality monitor's own counter takes precedence)
        MSDA.ms_deform_attn_forward(t["value"], t["shapes"], t["lsi"], t["loc"], t["aw"], 64)
        _lib.load().msda_debug_stats(None)
        _lib.set_option("fwd_variant", 0)
        print(f"general-path share: {cnt.item() / (2 * t['aw'].numel()):.4f}", flush=True)
    sweeps = []
    for s in args.set:
        k, v = s.split("=")
        sweeps.append((k, [int(x) for x in v.split(",")]))

    def run(label):
        _lib.profile_enable(4 * args.reps + 8)
        for _ in range(args.reps + 1):
            if args.only in ("fwd", "both"):
                MSDA.ms_deform_attn_forward(t["value"], t["shapes"], t["lsi"], t["loc"], t["aw"], 64)
            if args.only in ("bwd", "both"):
                MSDA.ms_deform_attn_backward(t["value"], t["shapes"], t["lsi"], t["loc"], t["aw"], t["grad_out"], 64)
        torch.cuda.synchronize()
        recs = _lib.profile_collect()
        _lib.profile_enable(0)
        by = {}
        for r in recs[2 if args.only == "both" else 1:]:
            by.setdefault((r["kind"], r["variant"]), []).append(r["kernel_ms"])
        print(label, " ".join(f"{k}/v{v}={1e3 * sum(ms) / len(ms):.1f}us" for (k, v), ms in sorted(by.items())), flush=True)

    def rec(i, label):
        if i == len(sweeps):
            run(label)
            return
        k, vals = sweeps[i]
        for v in vals:
            _lib.set_option(k, v)
            rec(i + 1, label + f"{k}={v} ")

    rec(0, f"{args.call}/{args.loc}: ")


if __name__ == "__main__":
    main()
