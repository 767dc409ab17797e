#!/usr/bin/env python3
"""Tuning aid: weight gradients of nn.Linear layers (dW = dY^T X, contraction over the tokens) through the convolution weight-gradient
kernel (a 1 x 1 convolution over a 1 x T image) against the library's transposed GEMM, bf16."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from richsem_amd.functions.linear import linear_wgrad_bf16   # noqa: E402


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


for T, cout, cin in [(t, o, i) for t in (44646, 2184) for o, i in ((256, 256), (384, 256), (512, 256), (256, 2048), (2048, 256))]:
    dy = torch.randn(T, cout, device="cuda").bfloat16()
    x = torch.randn(T, cin, device="cuda").bfloat16()
    want = dy.float().t() @ x.float()
    got = linear_wgrad_bf16(dy, x)
    err = float((got - want).abs().max() / want.abs().max())
    t1, t2 = timeit(lambda: linear_wgrad_bf16(dy, x)), timeit(lambda: dy.t() @ x)
    t3, t4 = timeit(lambda: linear_wgrad_bf16(dy, x, with_bias=True)), timeit(lambda: dy.sum(0, dtype=torch.float32))
    dwb, db = linear_wgrad_bf16(dy, x, with_bias=True)
    eb = float((db - dy.float().sum(0)).abs().max() / dy.float().sum(0).abs().max())
    fl = 2.0 * T * cout * cin
    print(f"dW ({cout} x {cin}), {T} tokens: kernel {t1:6.1f} us ({fl / t1 / 1e6:5.0f} TFLOP/s)  library bf16 {t2:6.1f} us   rel err vs fp32 {err:.1e};  with the bias sum {t3:6.1f} us (err {eb:.1e}; torch column sum alone {t4:5.1f} us)")
