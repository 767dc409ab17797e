"""CLIP's AttentionPool2d and RichSem's distillation targets built on it.

``AttentionPool2d`` mirrors the reference class (clip/model.py:58-91: same parameter names, so a CLIP ``visual.attnpool`` state dict
loads unchanged; same ``forward(x)`` for x of shape (K, C, H, W) -> (K, output_dim)).  On the GPU the attention is restructured
around its single query token (csrc/msda_attnpool.h): three C x C products per ROI instead of 2 (HW + 1) + 1, with the core
(scores, softmax, weighted token sum over tokens that are never materialised) in one HIP kernel ``msda_attnpool_core_*`` that reads
the ROIAlign kernel's output layout directly.  Forward only (the teacher is frozen, ``requires_grad_(False)`` in the reference).

``clip_box_targets`` mirrors models/richsem/richsem.py:745-761: ROIAlign of the ground-truth boxes on the frozen CLIP feature map
-> attention pool -> normalise -> logits against the normalised text embeddings, scaled by ``exp(logit_scale)``.
"""
import torch
from torch import nn

from .. import _lib
from ..roi import ROIAlign


class AttentionPool2d(nn.Module):
    def __init__(self, spacial_dim: int, embed_dim: int, num_heads: int, output_dim: int = None):
        super().__init__()
        self.positional_embedding = nn.Parameter(torch.randn(spacial_dim ** 2 + 1, embed_dim) / embed_dim ** 0.5)
        self.k_proj = nn.Linear(embed_dim, embed_dim)
        self.q_proj = nn.Linear(embed_dim, embed_dim)
        self.v_proj = nn.Linear(embed_dim, embed_dim)
        self.c_proj = nn.Linear(embed_dim, output_dim or embed_dim)
        self.num_heads = num_heads

    @torch.no_grad()
    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("Not implemented on the CPU")
        dt = x.dtype
        if dt not in (torch.float32, torch.float64):
            raise RuntimeError(f"AttentionPool2d: float32 / float64 input, got {dt}")
        K, C, Hh, Ww = x.shape
        T, H = Hh * Ww, self.num_heads
        assert self.positional_embedding.shape == (T + 1, C), "input resolution does not match the positional embedding"
        assert C % H == 0
        hd = C // H
        out_dim = self.c_proj.out_features
        if K == 0:
            return x.new_zeros(0, out_dim)
        feat = x.contiguous()
        pos = self.positional_embedding.to(dt).contiguous()
        x0 = feat.flatten(2).mean(dim=2) + pos[0]                                              # the query token (model.py:72)
        q = torch.addmm(self.q_proj.bias.to(dt), x0, self.q_proj.weight.to(dt).t()) * (hd ** -0.5)   # (K, C), scaled as F.mha does
        # u[k, h, :] = Wk_h^T q[k, h]: per head a (K, hd) x (hd, C) product
        u = torch.bmm(q.view(K, H, hd).transpose(0, 1), self.k_proj.weight.to(dt).view(H, hd, C)).transpose(0, 1).contiguous()
        z = torch.empty_like(u)
        fn = getattr(_lib.load(), "msda_attnpool_core_" + ("f32" if dt == torch.float32 else "f64"))
        with torch.cuda.device(x.device):
            _lib.check(fn(u.data_ptr(), feat.data_ptr(), pos.data_ptr(), K, H, C, T, z.data_ptr(),
                          torch.cuda.current_stream(x.device).cuda_stream))
        # o[k, h] = Wv_h z[k, h] + bv_h (the attention weights sum to one), then the output projection
        o = torch.bmm(z.transpose(0, 1), self.v_proj.weight.to(dt).view(H, hd, C).transpose(1, 2)).transpose(0, 1).reshape(K, C)
        o = o + self.v_proj.bias.to(dt)
        return torch.addmm(self.c_proj.bias.to(dt), o, self.c_proj.weight.to(dt).t())


@torch.no_grad()
def clip_box_targets(clip_features, targets, attnpool, text_embed, logit_scale, patch_size=32, grid_size=7):
    """richsem.py:745-761: per image the CLIP embedding (``clip_prompt``) and the text logits (``clip_logits``) of its ground-truth
    boxes.  clip_features (N, C, H/32, W/32) from the frozen teacher; targets: list of dicts with "boxes" (cxcywh, normalised),
    "size" (h, w) and "labels"; text_embed (classes, output_dim); logit_scale: the CLIP parameter (log of the scale).
    Returns (list of prompts, list of logits), split per image as the reference stores them into the targets."""
    dt, dev = clip_features.dtype, clip_features.device
    coord = torch.cat([t["boxes"] for t in targets]).to(dt)
    if len(coord):
        scale = torch.cat([t["size"][[1, 0, 1, 0]][None].expand(len(t["boxes"]), -1) for t in targets]).to(dt)
        cx, cy, w, h = coord.unbind(-1)
        xyxy = scale * torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)     # util/box_ops.py:9-13
        bidx = torch.cat([torch.full((len(t["boxes"]),), float(b), device=dev, dtype=dt) for b, t in enumerate(targets)])
        rois = torch.cat([bidx[:, None], xyxy], dim=-1)
        roi_features = ROIAlign(grid_size, 1.0 / patch_size, 0, aligned=True).forward(clip_features, rois)
        prompt = attnpool(roi_features)
        prompt = prompt / prompt.norm(dim=-1, keepdim=True)
    else:
        prompt = clip_features.new_zeros(0, text_embed.shape[-1])
    te = text_embed.to(dt)
    te = te / te.norm(dim=-1, keepdim=True)
    logits = (prompt @ te.t()) * torch.as_tensor(logit_scale, dtype=dt, device=dev).exp()
    sizes = [len(t["labels"]) for t in targets]
    return list(prompt.split(sizes, dim=0)), list(logits.split(sizes, dim=0))
