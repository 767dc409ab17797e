#!/usr/bin/env python3
"""Tuning aid: the top-k query selection kernel against torch.topk at the encoder's shape (2 x 22323 scores, k = 900)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from richsem_amd.dn import topk_indices   # noqa: E402

s = torch.randn(2, 22323, device="cuda")


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


print(f"msda_topk_f32 {timeit(lambda: topk_indices(s, 900)):7.1f} us   torch.topk {timeit(lambda: torch.topk(s, 900, dim=1)):7.1f} us")
