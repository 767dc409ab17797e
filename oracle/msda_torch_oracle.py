"""TEST INFRASTRUCTURE -- not part of the product path (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import it).

The reference's OWN CPU path for the op, restated: ``ms_deform_attn_core_pytorch`` (models/richsem/ops/functions/ms_deform_attn_func.py:41-61)
-- per level one ``F.grid_sample`` (bilinear, zero padding, align_corners=False) of the level's value map at ``2 * loc - 1``, then the
attention-weighted sum over levels x points.  It is what a user of the reference without the CUDA extension runs, so bench.py times it on
the host beside the C oracle (``cpu_baseline.grid_sample``).  PINNED: tests/test_oracle_golden.py checks it against the committed
fixtures, which were produced by the reference's function itself (tests/golden/make_golden.py).
"""
import torch
import torch.nn.functional as F


def core_grid_sample(value, spatial_shapes, sampling_locations, attention_weights):
    """value (N, S, M, D); spatial_shapes (L, 2) [(H, W)]; sampling_locations (N, Lq, M, L, P, 2) in [0, 1] (x, y);
    attention_weights (N, Lq, M, L, P) -> (N, Lq, M * D).  Differentiable (autograd) in value, locations and weights."""
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_locations.shape
    sizes = [int(h) * int(w) for h, w in spatial_shapes.tolist()]
    grids = 2 * sampling_locations - 1                                      # grid_sample's [-1, 1] convention
    per_level = []
    for lvl, (level_value, (H, W)) in enumerate(zip(value.split(sizes, dim=1), spatial_shapes.tolist())):
        maps = level_value.flatten(2).transpose(1, 2).reshape(N * M, D, int(H), int(W))            # (N*M, D, H, W)
        grid = grids[:, :, :, lvl].transpose(1, 2).flatten(0, 1)                                   # (N*M, Lq, P, 2)
        per_level.append(F.grid_sample(maps, grid, mode="bilinear", padding_mode="zeros", align_corners=False))   # (N*M, D, Lq, P)
    w = attention_weights.transpose(1, 2).reshape(N * M, 1, Lq, L * P)
    out = (torch.stack(per_level, dim=-2).flatten(-2) * w).sum(-1)          # (N*M, D, Lq)
    return out.view(N, M * D, Lq).transpose(1, 2).contiguous()


def forward_backward(value, spatial_shapes, sampling_locations, attention_weights, grad_out):
    """one forward + autograd backward; returns (out, grad_value, grad_loc, grad_aw)"""
    v, loc, aw = (t.detach().clone().requires_grad_(True) for t in (value, sampling_locations, attention_weights))
    out = core_grid_sample(v, spatial_shapes, loc, aw)
    out.backward(grad_out.reshape(out.shape))
    return out.detach(), v.grad, loc.grad, aw.grad
