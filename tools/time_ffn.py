#!/usr/bin/env python3
"""Tuning aid: the fused feed-forward kernel against its fp32 definition and against the same block as PyTorch bf16 ops.

    python tools/time_ffn.py [--tokens 44646] [--ffn 2048] [--reps 20]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from richsem_amd.functions import ffn_forward_bf16, pack_w2_bf16   # noqa: E402


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
    ap.add_argument("--tokens", type=int, default=44646)
    ap.add_argument("--ffn", type=int, default=2048)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    torch.manual_seed(0)
    T, D, Fh = args.tokens, 256, args.ffn
    dev = "cuda"
    x = torch.randn(T, D, device=dev).to(torch.bfloat16)
    w1 = (torch.randn(Fh, D, device=dev) * D ** -0.5).to(torch.bfloat16)
    w2 = (torch.randn(D, Fh, device=dev) * Fh ** -0.5).to(torch.bfloat16)
    b1, b2 = torch.randn(Fh, device=dev) * 0.1, torch.randn(D, device=dev) * 0.1
    gw, gb = 1 + 0.1 * torch.randn(D, device=dev), 0.1 * torch.randn(D, device=dev)
    w2p = pack_w2_bf16(w2)
    out = ffn_forward_bf16(x, w1, b1, w2p, b2, gw, gb)
    xf = x.float()
    ref = F.layer_norm(xf + F.linear(torch.relu(F.linear(xf, w1.float(), b1)), w2.float(), b2), (D,), gw, gb)
    err = (out.float() - ref).abs()
    print(f"tokens {T} d_ffn {Fh}: max |err| {err.max().item():.4f}  mean |err| {err.mean().item():.5f}  (|ref| mean {ref.abs().mean().item():.3f})")

    def torch_bf16():
        h = torch.relu(F.linear(x, w1, b1.to(torch.bfloat16)))
        return F.layer_norm(x + F.linear(h, w2, b2.to(torch.bfloat16)), (D,), gw.to(torch.bfloat16), gb.to(torch.bfloat16))
    e2 = (torch_bf16().float() - ref).abs()
    print(f"  PyTorch bf16 ops against the same fp32 definition: max |err| {e2.max().item():.4f}  mean |err| {e2.mean().item():.5f}")
    flop = 4.0 * T * D * Fh
    t_f = timeit(lambda: ffn_forward_bf16(x, w1, b1, w2p, b2, gw, gb), args.reps)
    t_t = timeit(torch_bf16, args.reps)
    print(f"  fused kernel {t_f:8.1f} us  {flop / t_f / 1e6:7.1f} TFLOP/s  ({flop / t_f / 1e6 / 2500:.3f} of 2.5 PFLOP/s dense bf16)")
    print(f"  PyTorch bf16 {t_t:8.1f} us  {flop / t_t / 1e6:7.1f} TFLOP/s")
    from richsem_amd import _lib
    buf = torch.zeros(((T + 191) // 192) * 8, dtype=torch.int64, device=dev)
    _lib.load().msda_ffn_debug_stamps(buf.data_ptr())
    ffn_forward_bf16(x, w1, b1, w2p, b2, gw, gb)
    torch.cuda.synchronize()
    _lib.load().msda_ffn_debug_stamps(None)
    st = buf.view(-1, 8).double().mean(0)
    names = ["wait for the weight tile", "barrier", "first product (48 MFMAs per tile)", "relu + conversion", "second product (48 MFMAs per tile)"]
    tot = float(st[:5].sum())
    for n, v in zip(names, st[:5].tolist()):
        print(f"    {n:36s} {v / (Fh // 32):8.0f} cycles per tile  {100 * v / tot:5.1f} %")


if __name__ == "__main__":
    main()
