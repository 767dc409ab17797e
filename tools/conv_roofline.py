#!/usr/bin/env python3
"""Where every convolution of ResNet-50 (2 x 800 x 1344, bf16 NHWC) stands against ITS OWN roofline: a convolution is bound by the
matrix pipe (2 * MACs / 2.5 PFLOP/s) or by HBM (input + output + weight bytes / 8 TB/s), whichever takes longer; the table gives the
measured time of the library's kernel (automatic tiling / ring choice), the two bounds, and the fraction of the binding one.

    python tools/conv_roofline.py [--reps 30] > profiles/r05_conv_roofline.md
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from richsem_amd.conv import ConvAffine   # noqa: E402

MFMA_PEAK, HBM_PEAK = 2.5e15, 8.0e12

# name, H, W, Cin, Cout, k, stride, pad, how many times the shape occurs in torchvision's ResNet-50 v1.5 (N = 2 images of 800 x 1344)
SHAPES = [
    ("stem 7x7 s2", 800, 1344, 3, 64, 7, 2, 3, 1),
    ("l1 1x1 64-64", 200, 336, 64, 64, 1, 1, 0, 1),
    ("l1 3x3 64", 200, 336, 64, 64, 3, 1, 1, 3),
    ("l1 1x1 64-256", 200, 336, 64, 256, 1, 1, 0, 4),          # 3 expansions + the downsample branch
    ("l1 1x1 256-64", 200, 336, 256, 64, 1, 1, 0, 2),
    ("l2 1x1 256-128", 200, 336, 256, 128, 1, 1, 0, 1),
    ("l2 3x3 s2 128", 200, 336, 128, 128, 3, 2, 1, 1),
    ("l2 1x1 s2 256-512", 200, 336, 256, 512, 1, 2, 0, 1),
    ("l2 3x3 128", 100, 168, 128, 128, 3, 1, 1, 3),
    ("l2 1x1 128-512", 100, 168, 128, 512, 1, 1, 0, 4),
    ("l2 1x1 512-128", 100, 168, 512, 128, 1, 1, 0, 3),
    ("l3 1x1 512-256", 100, 168, 512, 256, 1, 1, 0, 1),
    ("l3 3x3 s2 256", 100, 168, 256, 256, 3, 2, 1, 1),
    ("l3 1x1 s2 512-1024", 100, 168, 512, 1024, 1, 2, 0, 1),
    ("l3 3x3 256", 50, 84, 256, 256, 3, 1, 1, 5),
    ("l3 1x1 256-1024", 50, 84, 256, 1024, 1, 1, 0, 6),
    ("l3 1x1 1024-256", 50, 84, 1024, 256, 1, 1, 0, 5),
    ("l4 1x1 1024-512", 50, 84, 1024, 512, 1, 1, 0, 1),
    ("l4 3x3 s2 512", 50, 84, 512, 512, 3, 2, 1, 1),
    ("l4 1x1 s2 1024-2048", 50, 84, 1024, 2048, 1, 2, 0, 1),
    ("l4 3x3 512", 25, 42, 512, 512, 3, 1, 1, 2),
    ("l4 1x1 512-2048", 25, 42, 512, 2048, 1, 1, 0, 3),
    ("l4 1x1 2048-512", 25, 42, 2048, 512, 1, 1, 0, 2),
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
    return a.elapsed_time(b) / reps * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    args = ap.parse_args()
    torch.manual_seed(0)
    N = 2
    print("# Round 5: every ResNet-50 convolution against its own roofline (MI355X, 2 x 800 x 1344, bf16 NHWC, `tools/conv_roofline.py`)\n")
    print("bound = max(2 * MACs / 2.5 PFLOP/s, (input + output + weight bytes) / 8 TB/s); four distinct inputs are cycled so that a call does not find "
          "its input in the cache a network would not leave it in; affine + ReLU in the epilogue.\n")
    print("| convolution | x | measured us | TFLOP/s | GB/s | MFMA bound us | HBM bound us | binds | fraction of the binding roof |")
    print("|---|---|---|---|---|---|---|---|---|")
    tot_t = tot_bound = tot_flop = tot_bytes = 0.0
    for name, H, W, Cin, Cout, k, stride, pad, count in SHAPES:
        xs = [torch.randn(N, H, W, Cin, device="cuda").to(torch.bfloat16) for _ in range(4)]
        w = torch.randn(Cout, Cin, k, k, device="cuda") * (Cin * k * k) ** -0.5
        conv = ConvAffine(w, None, None, stride, pad, relu=True)
        state = {"i": 0}

        def run():
            state["i"] = (state["i"] + 1) & 3
            conv(xs[state["i"]])
        t = timeit(run, args.reps)
        Ho, Wo = conv.out_hw(H, W)
        flop = 2.0 * N * Ho * Wo * Cout * Cin * k * k
        nbytes = 2.0 * (N * H * W * Cin + N * Ho * Wo * Cout + Cout * Cin * k * k)
        t_m, t_h = flop / MFMA_PEAK, nbytes / HBM_PEAK
        bound = max(t_m, t_h)
        print(f"| {name} | {count} | {t * 1e6:.1f} | {flop / t / 1e12:.0f} | {nbytes / t / 1e9:.0f} | {t_m * 1e6:.1f} | {t_h * 1e6:.1f} | "
              f"{'MFMA' if t_m >= t_h else 'HBM'} | {bound / t:.2f} |", flush=True)
        tot_t += count * t
        tot_bound += count * bound
        tot_flop += count * flop
        tot_bytes += count * nbytes
    print(f"\nWhole backbone (shapes x their counts): measured {tot_t * 1e3:.2f} ms for {tot_flop / 1e9:.0f} GFLOP and {tot_bytes / 1e9:.2f} GB of tensor traffic; "
          f"sum of the per-convolution bounds {tot_bound * 1e3:.3f} ms -> {tot_bound / tot_t:.2f} of its own roofline; "
          f"{tot_flop / tot_t / 1e12:.0f} TFLOP/s = {tot_flop / tot_t / MFMA_PEAK:.3f} of the MFMA peak, {tot_bytes / tot_t / 1e9:.0f} GB/s = {tot_bytes / tot_t / HBM_PEAK:.3f} of the HBM peak.")


if __name__ == "__main__":
    main()
