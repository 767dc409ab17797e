"""Eight forward + backward steps of one encoder layer on the bf16 path (for rocprofv3 traces: tools/trace_table.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from richsem_amd import workload as W
from richsem_amd.modules import DeformableTransformerEncoderLayer, get_reference_points
torch.manual_seed(0)
call = W.call_E(2); shapes, lsi = W.level_tensors(call, "cuda")
layer = DeformableTransformerEncoderLayer(256, 2048, dropout=0.0, n_levels=4, n_heads=8, n_points=4).cuda()
with torch.no_grad():
    layer.self_attn.sampling_offsets.weight.normal_(0, 0.01); layer.self_attn.attention_weights.weight.normal_(0, 0.1)
src = torch.randn(call.N, call.S, 256, device="cuda").bfloat16().requires_grad_(True); pos = (0.1 * torch.randn(call.N, call.S, 256, device="cuda")).bfloat16()
ref = get_reference_points(shapes.tolist(), torch.ones(call.N, call.L, 2, device="cuda"), "cuda"); go = torch.randn(call.N, call.S, 256, device="cuda").bfloat16()
for _ in range(8):
    for q in layer.parameters(): q.grad = None
    src.grad = None
    layer(src, pos, ref, shapes, lsi, None).backward(go)
torch.cuda.synchronize()
