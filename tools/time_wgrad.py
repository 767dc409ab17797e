#!/usr/bin/env python3
"""Tuning aid: the weight-gradient kernel (csrc/conv_wgrad.hip) and the input-gradient call per convolution shape of ResNet-50's trained
stages at 2 x 800 x 1344, against MIOpen (aten::convolution_backward, bf16 channels-last).

    python tools/time_wgrad.py [--reps 10]
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from richsem_amd import _lib   # noqa: E402
from richsem_amd.conv import _pack_form   # noqa: E402

SHAPES = [  # name, H, W (input), Cin, Cout, k, stride, pad    (N = 2)
    ("l2 1x1 256-128", 200, 336, 256, 128, 1, 1, 0),
    ("l2 3x3 s2 128", 200, 336, 128, 128, 3, 2, 1),
    ("l2 1x1 128-512", 100, 168, 128, 512, 1, 1, 0),
    ("l2 1x1 512-128", 100, 168, 512, 128, 1, 1, 0),
    ("l2 3x3 128", 100, 168, 128, 128, 3, 1, 1),
    ("l3 1x1 512-256", 100, 168, 512, 256, 1, 1, 0),
    ("l3 3x3 s2 256", 100, 168, 256, 256, 3, 2, 1),
    ("l3 1x1 256-1024", 50, 84, 256, 1024, 1, 1, 0),
    ("l3 1x1 1024-256", 50, 84, 1024, 256, 1, 1, 0),
    ("l3 3x3 256", 50, 84, 256, 256, 3, 1, 1),
    ("l4 3x3 512", 25, 42, 512, 512, 3, 1, 1),
    ("l4 1x1 512-2048", 25, 42, 512, 2048, 1, 1, 0),
    ("l4 1x1 2048-512", 25, 42, 2048, 512, 1, 1, 0),
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    torch.manual_seed(0)
    L = _lib.load()
    N = 2
    st = torch.cuda.current_stream().cuda_stream
    print(f"{'shape':18s} {'GFLOP':>6s} | {'wgrad us':>8s} {'TFLOP/s':>7s} {'MIOpen':>7s} | {'dgrad us':>8s} {'TFLOP/s':>7s} {'MIOpen':>7s}")
    tot = [0, 0, 0, 0]
    for name, H, W, Cin, Cout, k, stride, pad in SHAPES:
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        x = torch.randn(N, H, W, Cin, device="cuda").to(torch.bfloat16)
        dz = torch.randn(N, Ho, Wo, Cout, device="cuda").to(torch.bfloat16)
        w = torch.randn(Cout, Cin, k, k, device="cuda") * (Cin * k * k) ** -0.5
        scale = torch.ones(Cout, device="cuda")
        dw = torch.empty(Cout, k, k, Cin, device="cuda")
        nb = ctypes.c_int64(0)
        _lib.check(L.msda_conv_wgrad_workspace_bytes(N, H, W, Cin, Cout, k, k, stride, pad, ctypes.byref(nb)))
        ws = torch.empty(max(nb.value // 4, 4), device="cuda")
        pk = _pack_form(w, scale, True)
        dx = torch.empty_like(x)

        def wgrad():
            _lib.check(L.msda_conv_wgrad_bf16(dz.data_ptr(), x.data_ptr(), N, H, W, Cin, Cout, k, k, stride, pad, dw.data_ptr(), None, None, 0, ws.data_ptr(), st))

        def dgrad():
            _lib.check(L.msda_conv_dgrad_bf16(dz.data_ptr(), pk.data_ptr(), N, Ho, Wo, Cout, Cin, k, k, stride, pad, H, W, dx.data_ptr(), st))

        w16 = w.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        xn, dzn = x.permute(0, 3, 1, 2), dz.permute(0, 3, 1, 2)

        def lib(mask):
            return torch.ops.aten.convolution_backward(dzn, xn, w16, None, [stride, stride], [pad, pad], [1, 1], False, [0, 0], 1, mask)

        flop = 2.0 * N * Ho * Wo * Cout * Cin * k * k
        tw, td = timeit(wgrad, args.reps), timeit(dgrad, args.reps)
        tlw, tld = timeit(lambda: lib([False, True, False]), args.reps), timeit(lambda: lib([True, False, False]), args.reps)
        print(f"{name:18s} {flop / 1e9:6.2f} | {tw:8.1f} {flop / tw / 1e6:7.1f} {tlw:7.1f} | {td:8.1f} {flop / td / 1e6:7.1f} {tld:7.1f}   (split workspace {nb.value / 1e6:.1f} MB)", flush=True)
        for i, v in enumerate((tw, tlw, td, tld)):
            tot[i] += v
    print(f"{'sum':18s}        | {tot[0]:8.1f}         {tot[1]:7.1f} | {tot[2]:8.1f}         {tot[3]:7.1f}")


if __name__ == "__main__":
    main()
