#!/usr/bin/env python3
"""Whole-layer timing: DeformableTransformerDecoderLayer (reference models/richsem/deformable_transformer.py:883-1066, ['sa', 'ca',
'ffn']) forward + backward at the training shape (1092 queries x 2 images against 22323 memory tokens): the bf16 path on the library's
kernels, the fp32 op sequence around the operator, and the six-layer TransformerDecoder (stacked value projection).

    python tools/time_decoder_layer.py [--reps 10] [--only bf16]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from richsem_amd import workload as W   # noqa: E402
from richsem_amd.capture import quiet_gc   # noqa: E402
from richsem_amd.modules import MLP, DeformableTransformerDecoderLayer, TransformerDecoder   # noqa: E402


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


def timeit_graph(fn, reps):
    """the same work captured once into a HIP graph and replayed: GPU time without the host's launch gaps"""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with quiet_gc(), torch.cuda.graph(g, stream=side):      # (the warmed stream: the library's workspaces are per stream)
        fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only", default="")
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
    amask = torch.zeros(nq, nq, dtype=torch.bool, device="cuda")
    amask[192:, :192] = True
    go = torch.randn_like(tgt)

    def run_layer(dt, fused_attn=True):
        layer.cross_attn.fused = fused_attn
        t, m = tgt.to(dt).clone().requires_grad_(True), memory.to(dt).clone().requires_grad_(True)

        def step():
            for q in layer.parameters():
                q.grad = None
            layer(tgt=t, tgt_query_pos=qpos.to(dt), tgt_reference_points=refp, memory=m, memory_level_start_index=lsi,
                  memory_spatial_shapes=shapes, self_attn_mask=amask).backward(go.to(dt))
        return (timeit(step, args.reps), timeit_graph(step, args.reps)) if dt == torch.bfloat16 else timeit(step, args.reps)

    t16, g16 = run_layer(torch.bfloat16)
    msg = (f"decoder layer forward + backward, {nq} queries x {bs} images, {call.S} memory tokens: bf16 on the library's kernels {t16:.0f} us "
           f"eager, {g16:.0f} us replayed as a HIP graph")
    if args.only != "bf16":
        msg += f"; fp32 with the fused attention module {run_layer(torch.float32):.0f} us; fp32 op-by-op {run_layer(torch.float32, False):.0f} us"
    print(msg, flush=True)

    dec = TransformerDecoder(layer, 6, torch.nn.LayerNorm(256), d_model=256).cuda()
    dec.bbox_embed = torch.nn.ModuleList([MLP(256, 256, 4, 3) for _ in range(6)]).cuda()
    refu = torch.randn(nq, bs, 4, device="cuda")
    vr = torch.ones(bs, 4, 2, device="cuda")

    def run_stack(dt):
        t, m = tgt.to(dt).clone().requires_grad_(True), memory.to(dt).clone().requires_grad_(True)

        def step():
            for q in dec.parameters():
                q.grad = None
            hs, refs = dec(tgt=t, memory=m, tgt_mask=amask, refpoints_unsigmoid=refu, level_start_index=lsi, spatial_shapes=shapes, valid_ratios=vr)
            (torch.stack(hs).float().square().mean() + torch.stack(refs).square().mean()).backward()
        return (timeit(step, args.reps), timeit_graph(step, args.reps)) if dt == torch.bfloat16 else timeit(step, args.reps)

    s16, sg16 = run_stack(torch.bfloat16)
    msg = (f"six-layer decoder forward + backward: bf16 on the library's kernels {s16:.0f} us eager ({s16 / 6:.0f} us per layer), "
           f"{sg16:.0f} us as a HIP graph ({sg16 / 6:.0f} us per layer)")
    if args.only != "bf16":
        msg += f"; fp32 {run_stack(torch.float32):.0f} us"
    print(msg, flush=True)


if __name__ == "__main__":
    main()
