#!/usr/bin/env python3
"""Tuning aid: the MFMA convolution (csrc/conv_mfma.hip) at the convolution shapes of ResNet-50 / CLIP-RN50 on an 800 x 1344 batch of 2,
against torch.nn.functional.conv2d (MIOpen) in bf16 channels-last with the affine + ReLU as separate ops.

    python tools/time_conv.py [--reps 10]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from richsem_amd.conv import ConvAffine   # noqa: E402

SHAPES = [  # name, H, W, Cin, Cout, k, stride, pad   (N = 2; torchvision ResNet-50 v1.5 at 800 x 1344)
    ("stem 7x7 s2", 800, 1344, 3, 64, 7, 2, 3),
    ("l1 1x1 64-64", 200, 336, 64, 64, 1, 1, 0),
    ("l1 3x3 64", 200, 336, 64, 64, 3, 1, 1),
    ("l1 1x1 64-256", 200, 336, 64, 256, 1, 1, 0),
    ("l1 1x1 256-64", 200, 336, 256, 64, 1, 1, 0),
    ("l2 3x3 128", 100, 168, 128, 128, 3, 1, 1),
    ("l2 1x1 128-512", 100, 168, 128, 512, 1, 1, 0),
    ("l2 1x1 512-128", 100, 168, 512, 128, 1, 1, 0),
    ("l3 3x3 256", 50, 84, 256, 256, 3, 1, 1),
    ("l3 1x1 256-1024", 50, 84, 256, 1024, 1, 1, 0),
    ("l3 1x1 1024-256", 50, 84, 1024, 256, 1, 1, 0),
    ("l4 3x3 512", 25, 42, 512, 512, 3, 1, 1),
    ("l4 1x1 512-2048", 25, 42, 512, 2048, 1, 1, 0),
    ("l4 1x1 2048-512", 25, 42, 2048, 512, 1, 1, 0),
    ("l3 3x3 s2 256", 100, 168, 256, 256, 3, 2, 1),
]


def timeit(fn, reps):
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


def sweep(args):
    from richsem_amd.conv import set_tiling
    torch.manual_seed(0)
    N = 2
    for name, H, W, Cin, Cout, k, stride, pad in SHAPES:
        x = torch.randn(N, H, W, Cin, device="cuda").to(torch.bfloat16)
        w = torch.randn(Cout, Cin, k, k, device="cuda") * (Cin * k * k) ** -0.5
        conv = ConvAffine(w, None, None, stride, pad, relu=True)
        set_tiling(0, 0)
        auto = timeit(lambda: conv(x), args.reps)
        res = []
        for ct in (16, 8, 4, 2, 1):
            if (Cout // 16) % ct or (ct < 4 and (Cout // 16) % 4 == 0):
                continue
            for pt in (3, 2, 1):
                set_tiling(ct, pt)
                res.append((timeit(lambda: conv(x), args.reps), ct, pt))
        set_tiling(0, 0)
        res.sort()
        print(f"{name:18s} auto {auto:7.1f} us | " + "  ".join(f"({ct:2d},{pt}) {t:6.1f}" for t, ct, pt in res[:6]), flush=True)


def ring_sweep(args):
    """conv_ring_kernel (operands prefetched through LDS rings) against conv_fwd_kernel per shape: ring slots x tile shapes.  The library
    entry is called directly on preallocated buffers (no workspace: no k split), 3 + reps launches between two events."""
    import ctypes
    from richsem_amd import _lib
    from richsem_amd.conv import set_ring, set_tiling, _pack_form
    L = _lib.load()
    torch.manual_seed(0)
    N = 2
    st = torch.cuda.current_stream().cuda_stream
    tot = {}
    shapes = [(n, H, W, ci, co, k, s_, p_, False) for n, H, W, ci, co, k, s_, p_ in SHAPES if ci % 64 == 0]
    shapes += [("dgrad l3 3x3 s2", 100, 168, 256, 256, 3, 2, 1, True), ("dgrad l4 3x3 s2", 50, 84, 512, 512, 3, 2, 1, True),
               ("dgrad l3 down s2", 100, 168, 512, 1024, 1, 2, 0, True), ("dgrad l4 down s2", 50, 84, 1024, 2048, 1, 2, 0, True)]
    only = [n.strip() for n in args.only.split(",") if n.strip()]
    for name, H, W, Cin, Cout, k, stride, pad, dgrad in shapes:
        if only and name not in only:
            continue
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        w = torch.randn(Cout, Cin, k, k, device="cuda") * (Cin * k * k) ** -0.5
        one = torch.ones(Cout, device="cuda")
        zero = torch.zeros(Cout, device="cuda")
        packed = _pack_form(w, one, dgrad)
        if dgrad:
            src = torch.randn(N, Ho, Wo, Cout, device="cuda").to(torch.bfloat16)
            dst = torch.empty(N, H, W, Cin, device="cuda", dtype=torch.bfloat16)
            call = lambda: L.msda_conv_dgrad_ws_bf16(src.data_ptr(), packed.data_ptr(), N, Ho, Wo, Cout, Cin, k, k, stride, pad, H, W, dst.data_ptr(), None, st)
            couts = Cin
        else:
            src = torch.randn(N, H, W, Cin, device="cuda").to(torch.bfloat16)
            dst = torch.empty(N, Ho, Wo, Cout, device="cuda", dtype=torch.bfloat16)
            call = lambda: L.msda_conv_forward_ws_bf16(src.data_ptr(), packed.data_ptr(), one.data_ptr(), zero.data_ptr(), None, N, H, W, Cin, Cout, k, k,
                                                        stride, pad, 1, dst.data_ptr(), None, st)
            couts = Cout
        set_tiling(0, 0)
        set_ring(-1)
        auto = timeit(call, args.reps)
        line = []
        for slots in (-1, 3, 4, 6):
            res = []
            for ct in (16, 8, 4, 2):
                if (couts // 16) % ct or (ct < 4 and (couts // 16) % 4 == 0):
                    continue
                for pt in (2, 1):
                    set_tiling(ct, pt)
                    set_ring(slots)
                    res.append((timeit(call, args.reps), ct, pt))
            res.sort()
            tot[slots] = tot.get(slots, 0.0) + res[0][0]
            line.append(f"ring {slots:2d}: " + " ".join(f"({ct:2d},{pt}) {t:5.1f}" for t, ct, pt in res[:3]))
        set_tiling(0, 0)
        set_ring(0)
        tot["auto"] = tot.get("auto", 0.0) + auto
        print(f"{name:18s} auto {auto:6.1f} us | " + " | ".join(line), flush=True)
    print("sums of the best per shape:", {k: round(v, 1) for k, v in tot.items()})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="", help="with --ring: comma-separated shape names")
    ap.add_argument("--ring", action="store_true", help="conv_ring_kernel against conv_fwd_kernel: ring slots x tile shapes per convolution")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--sweep", action="store_true", help="time every tile shape per convolution (msda_conv_set_tiling)")
    args = ap.parse_args()
    if args.sweep:
        return sweep(args)
    if args.ring:
        return ring_sweep(args)
    torch.manual_seed(0)
    N = 2
    print(f"{'shape':18s} {'GFLOP':>7s} {'mfma us':>8s} {'TFLOP/s':>8s} {'of peak':>8s} {'MIOpen bf16 us':>15s} {'(+bn+relu)':>11s}")
    tot = [0.0, 0.0, 0.0, 0.0]
    for name, H, W, Cin, Cout, k, stride, pad in SHAPES:
        x = torch.randn(N, H, W, Cin, device="cuda").to(torch.bfloat16)
        w = torch.randn(Cout, Cin, k, k, device="cuda") * (Cin * k * k) ** -0.5
        scale, shift = 1 + 0.1 * torch.randn(Cout, device="cuda"), 0.1 * torch.randn(Cout, device="cuda")
        conv = ConvAffine(w, scale, shift, stride, pad, relu=True)
        Ho, Wo = conv.out_hw(H, W)
        flop = 2.0 * N * Ho * Wo * Cout * Cin * k * k
        t = timeit(lambda: conv(x), args.reps)
        xc = x.permute(0, 3, 1, 2)     # NCHW view of channels-last memory
        wc = w.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        s16, b16 = scale.to(torch.bfloat16)[None, :, None, None], shift.to(torch.bfloat16)[None, :, None, None]
        t_conv = timeit(lambda: F.conv2d(xc, wc, stride=stride, padding=pad), args.reps)
        t_full = timeit(lambda: torch.relu(F.conv2d(xc, wc, stride=stride, padding=pad) * s16 + b16), args.reps)
        print(f"{name:18s} {flop / 1e9:7.2f} {t:8.1f} {flop / t / 1e6:8.1f} {flop / t / 1e6 / 2500:8.3f} {t_conv:15.1f} {t_full:11.1f}")
        tot[0] += flop; tot[1] += t; tot[2] += t_conv; tot[3] += t_full
    print(f"{'sum':18s} {tot[0] / 1e9:7.2f} {tot[1]:8.1f} {tot[0] / tot[1] / 1e6:8.1f} {tot[0] / tot[1] / 1e6 / 2500:8.3f} {tot[2]:15.1f} {tot[3]:11.1f}")


if __name__ == "__main__":
    main()
