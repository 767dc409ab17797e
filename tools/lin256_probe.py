#!/usr/bin/env python3
"""Tuning aid: csrc/lin256_mfma.hip (the feed-forward backward's two token-parallel products) against the library's bf16 ops."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from richsem_amd.functions.linear import lin256, lin256_pack   # noqa: E402


def timeit(fn, reps=10):
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


T, N = 44646, 2048
x = torch.randn(T, 256, device="cuda").bfloat16()
w = (torch.randn(N, 256, device="cuda") / 16).bfloat16()
b = torch.randn(N, device="cuda")
wp = lin256_pack(w)
h = lin256(x, wp, b, relu=True)
ref = torch.relu(x.float() @ w.float().t() + b)
print("relu epilogue: max err", float((h.float() - ref).abs().max()), "of", float(ref.abs().max()))
gh = lin256(x, wp, relu_mask=h)
refm = (x.float() @ w.float().t()) * (h > 0)
print("mask epilogue: max err", float((gh.float() - refm).abs().max()), "of", float(refm.abs().max()))
fl = 2.0 * T * 256 * N
t1 = timeit(lambda: lin256(x, wp, b, relu=True))
t2 = timeit(lambda: torch.relu(torch.addmm(b.bfloat16(), x, w.t())))
t3 = timeit(lambda: lin256(x, wp, relu_mask=h))
t4 = timeit(lambda: torch.ops.aten.threshold_backward(x @ w.t(), h, 0))
print(f"h = relu(x W^T + b): kernel {t1:.1f} us ({fl / t1 / 1e6:.0f} TFLOP/s), library ops {t2:.1f} us;  gh = (x W^T) * (h > 0): kernel {t3:.1f} us, library ops {t4:.1f} us")

from richsem_amd.functions.linear import lin256_f32, lin256_f32_pack   # noqa: E402
xf = torch.randn(T, 256, device="cuda")
for n in (256, 384):
    wf = torch.randn(n, 256, device="cuda") / 16
    bf = torch.randn(n, device="cuda")
    pk = lin256_f32_pack(wf)
    got = lin256_f32(xf, pk, n, bf)
    want = (xf.double() @ wf.double().t() + bf.double())
    lib = torch.nn.functional.linear(xf, wf, bf)
    e1, e2 = float((got.double() - want).abs().max() / want.abs().max()), float((lib.double() - want).abs().max() / want.abs().max())
    t1, t2 = timeit(lambda: lin256_f32(xf, pk, n, bf)), timeit(lambda: torch.nn.functional.linear(xf, wf, bf))
    print(f"fp32 linear 256 -> {n}: split-bf16 kernel {t1:.1f} us (err {e1:.1e}), library fp32 {t2:.1f} us (err {e2:.1e})")
