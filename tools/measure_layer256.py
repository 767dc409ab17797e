#!/usr/bin/env python3
"""Tuning / evidence aid: the errors of the bf16 layer paths against the d_model = 256 reference-class fixtures (what the bounds of
tests/test_gpu_layers.py: BF16_BOUNDS are twice of)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_layers as T   # noqa: E402

for kind in ("encoder", "decoder", "stack"):
    for dtype, act in ((torch.float64, torch.float64), (torch.float32, torch.float32), (torch.float32, torch.bfloat16)):
        res, m, gr = T._run256(kind, dtype, act)
        print(f"== {kind} params {dtype} activations {act}")
        for k, (got, want) in res.items():
            print(f"   {k:18s} max {T._rel(got.double(), want):.3e}  mean {T._mean_rel(got, want):.3e}")
        pe = T._errs_params(m, gr)
        wmax = max(pe.items(), key=lambda kv: kv[1][0])
        wmean = max(pe.items(), key=lambda kv: kv[1][1])
        print(f"   params ({len(pe)}): worst max {wmax[1][0]:.3e} ({wmax[0]}), worst mean {wmean[1][1]:.3e} ({wmean[0]})")
        if act == torch.bfloat16:
            for k, e in sorted(pe.items()):
                print(f"      {k:48s} max {e[0]:.3e} mean {e[1]:.3e}")
# yardstick: the same bf16 activations through PyTorch's own ops (the mirrors' op-by-op path: fused = False)
for kind in ("encoder", "decoder"):
    orig = T._build256

    def unfused(kind_, dtype_):
        m = orig(kind_, dtype_)
        for x in m.modules():
            if hasattr(x, "fused"):
                x.fused = False
            if hasattr(x, "fused_ffn"):
                x.fused_ffn = False
        return m
    T._build256 = unfused
    res, m, gr = T._run256(kind, torch.bfloat16, torch.bfloat16)      # (parameters in bf16 too: PyTorch's ops do not mix dtypes)
    T._build256 = orig
    print(f"== {kind} bf16 activations, PyTorch ops (fused = False)")
    for k, (got, want) in res.items():
        print(f"   {k:18s} max {T._rel(got.double(), want):.3e}  mean {T._mean_rel(got, want):.3e}")
    pe = T._errs_params(m, gr)
    for k, e in sorted(pe.items()):
        print(f"      {k:48s} max {e[0]:.3e} mean {e[1]:.3e}")
