#!/usr/bin/env python3
"""Whole-layer timing: one DeformableTransformerEncoderLayer (reference models/richsem/deformable_transformer.py:825-881) forward +
backward at the training shape (2 x 22323 tokens, d_model 256, d_ffn 2048): the library's bf16 path (fused MSDeformAttn module + one-kernel
feed-forward block with its backward kernels) against the same layer with every fused piece switched off (bf16 op-by-op: PyTorch ops
around the operator's bf16 kernels) and against the fp32 op-by-op layer (the reference's precision and op sequence).

    python tools/time_encoder_layer.py [--reps 5]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from richsem_amd import workload as W   # noqa: E402
from richsem_amd.modules import DeformableTransformerEncoderLayer, get_reference_points   # noqa: E402


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
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    torch.manual_seed(0)
    call = W.call_E(2)
    shapes, lsi = W.level_tensors(call, "cuda")
    layer = DeformableTransformerEncoderLayer(256, 2048, dropout=0.0, n_levels=4, n_heads=8, n_points=4).cuda()
    with torch.no_grad():
        layer.self_attn.sampling_offsets.weight.normal_(0, 0.01)
        layer.self_attn.attention_weights.weight.normal_(0, 0.1)
    src = torch.randn(call.N, call.S, 256, device="cuda")
    pos = 0.1 * torch.randn_like(src)
    valid = torch.ones(call.N, call.L, 2, device="cuda")
    ref = get_reference_points(shapes.tolist(), valid, "cuda")
    go = torch.randn_like(src)

    def run(dt, fused):
        layer.self_attn.fused = fused
        layer.fused_ffn = fused
        s, p, g = src.to(dt).requires_grad_(True), pos.to(dt), go.to(dt)

        def step():
            for q in layer.parameters():
                q.grad = None
            s.grad = None
            layer(s, p, ref, shapes, lsi, None).backward(g)
        return timeit(step, args.reps)

    t_fused = run(torch.bfloat16, True)
    t_ops16 = run(torch.bfloat16, False) if False else None        # (the op-by-op attention module has no bf16 operator binding)
    t_fused32 = run(torch.float32, True)
    t_ops32 = run(torch.float32, False)
    print(f"encoder layer forward + backward, 2 x {call.S} tokens: library bf16 path {t_fused:.0f} us;  fp32 with the fused attention module "
          f"{t_fused32:.0f} us;  fp32 op-by-op (the reference's op sequence around the operator) {t_ops32:.0f} us")


if __name__ == "__main__":
    main()
