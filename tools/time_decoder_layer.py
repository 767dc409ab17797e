#!/usr/bin/env python3
"""Whole-layer timing: one DeformableTransformerDecoderLayer (reference models/richsem/deformable_transformer.py:883-1066, ['sa', 'ca',
'ffn']) forward + backward at the training shape (1092 queries x 2 images against 22323 memory tokens), fp32: with the fused attention
module and op by op (the reference's sequence around the operator).

    python tools/time_decoder_layer.py [--reps 10]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from richsem_amd import workload as W   # noqa: E402
from richsem_amd.modules import DeformableTransformerDecoderLayer   # noqa: E402


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
    call = W.call_Dd(2)
    shapes, lsi = W.level_tensors(call, "cuda")
    nq, bs = call.Lq, call.N
    layer = DeformableTransformerDecoderLayer(256, 2048, dropout=0.0, n_levels=4, n_heads=8, n_points=4).cuda()
    with torch.no_grad():
        layer.cross_attn.sampling_offsets.weight.normal_(0, 0.01)
        layer.cross_attn.attention_weights.weight.normal_(0, 0.1)
    tgt, qpos, memory = torch.randn(nq, bs, 256, device="cuda"), 0.1 * torch.randn(nq, bs, 256, device="cuda"), torch.randn(call.S, bs, 256, device="cuda")
    refp = torch.rand(nq, bs, 4, 4, device="cuda") * 0.5 + 0.2
    go = torch.randn_like(tgt)

    def run(fused):
        layer.cross_attn.fused = fused
        t, m = tgt.clone().requires_grad_(True), memory.clone().requires_grad_(True)

        def step():
            for q in layer.parameters():
                q.grad = None
            layer(t, qpos, None, None, refp, m, None, lsi, shapes).backward(go)
        return timeit(step, args.reps)

    t32f = run(True)
    t32 = run(False)
    print(f"decoder layer forward + backward, {nq} queries x {bs} images, {call.S} memory tokens: fp32 with the fused attention module {t32f:.0f} us; "
          f"fp32 op-by-op {t32:.0f} us")


if __name__ == "__main__":
    main()
