#!/usr/bin/env python3
"""msda_narrow_linear_backward_bf16 (the box heads' 256 -> 4 layer) called directly: us per call (two memsets + dx kernel + dw kernel)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from richsem_amd import _lib      # noqa: E402

L = _lib.load()
for T in (2184, 44646):
    x = torch.randn(T, 256, device="cuda").to(torch.bfloat16)
    dy = torch.randn(T, 4, device="cuda").to(torch.bfloat16)
    w = torch.randn(4, 256, device="cuda")
    dx = torch.empty_like(x)
    dwb = torch.empty(4 * 256 + 8, device="cuda")
    st = _lib.raw_stream(x.device)

    def call(with_dx):
        _lib.check(L.msda_narrow_linear_backward_bf16(dy.data_ptr(), x.data_ptr(), w.data_ptr(), T, 4, dx.data_ptr() if with_dx else None, dwb.data_ptr(),
                                                      dwb[1024:].data_ptr(), st))
    for with_dx in (True, False):
        for _ in range(3):
            call(with_dx)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(100):
            call(with_dx)
        b.record()
        torch.cuda.synchronize()
        print(f"{T:6d} tokens, {'dx + ' if with_dx else ''}dw + db: {a.elapsed_time(b) * 10:.1f} us per call", flush=True)
    want = dy.float().t() @ x.float()
    assert float((dwb[:1024].view(4, 256) - want).abs().max()) <= 1e-4 * float(want.abs().max())
