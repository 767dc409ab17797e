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
        d = self._derived(dt)
        x0 = feat.flatten(2).mean(dim=2) + d["pos"][0]                                         # the query token (model.py:72)
        q = torch.addmm(d["bq"], x0, d["wq_t"])                                                # (K, C), scaled as F.mha does
        # u[h, k, :] = Wk_h^T q[k, h]: one batched product over the heads, (H, K, hd) x (H, hd, C) -> (H, K, C)
        u = torch.bmm(q.view(K, H, hd).transpose(0, 1), d["wk_h"])
        z = torch.empty_like(u)
        spos = torch.mm(u.view(H * K, C), d["pos_t"])                                          # u . pos_t for every (head, ROI): (H K, T + 1)
        fn = getattr(_lib.load(), "msda_attnpool_core_" + ("f32" if dt == torch.float32 else "f64"))
        with _lib.on_device(x.device):
            _lib.check(fn(u.data_ptr(), feat.data_ptr(), d["pos"].data_ptr(), spos.data_ptr(), K, H, C, T, 1, z.data_ptr(),
                          _lib.raw_stream(x.device)))
        # o[k, h] = Wv_h z[h, k] (+ bv_h, folded into the output bias: the attention weights sum to one), then the output projection
        o = torch.bmm(z, d["wv_ht"]).transpose(0, 1).reshape(K, C)
        return torch.addmm(d["bc"], o, d["wc_t"])

    def _derived(self, dt):
        """per-dtype forms of the (frozen) parameters, rebuilt when a parameter has been modified: scaled query projection, per-head
        views of the key / value projections, the output bias with the value bias folded in"""
        ps = (self.positional_embedding, self.q_proj.weight, self.q_proj.bias, self.k_proj.weight, self.v_proj.weight, self.v_proj.bias,
              self.c_proj.weight, self.c_proj.bias)
        ver = (dt,) + tuple((p.data_ptr(), p._version) for p in ps)
        if getattr(self, "_derived_ver", None) != ver:
            C, H = self.q_proj.weight.shape[0], self.num_heads
            hd = C // H
            scale = hd ** -0.5
            wc = self.c_proj.weight.detach().to(dt)
            self._derived_cache = {
                "pos": self.positional_embedding.detach().to(dt).contiguous(),
                "pos_t": self.positional_embedding.detach().to(dt).t().contiguous(),
                "wq_t": (self.q_proj.weight.detach().to(dt) * scale).t().contiguous(),
                "bq": (self.q_proj.bias.detach().to(dt) * scale).contiguous(),
                "wk_h": self.k_proj.weight.detach().to(dt).view(H, hd, C).contiguous(),
                "wv_ht": self.v_proj.weight.detach().to(dt).view(H, hd, C).transpose(1, 2).contiguous(),
                "wc_t": wc.t().contiguous(),
                "bc": (self.c_proj.bias.detach().to(dt) + wc @ self.v_proj.bias.detach().to(dt)).contiguous(),
            }
            self._derived_ver = ver
        return self._derived_cache


@torch.no_grad()
def clip_box_targets(clip_features, targets, attnpool, text_embed, logit_scale, patch_size=32, grid_size=7):
    """richsem.py:745-761: per image the CLIP embedding (``clip_prompt``) and the text logits (``clip_logits``) of its ground-truth
    boxes.  clip_features (N, C, H/32, W/32) from the frozen teacher; targets: list of dicts with "boxes" (cxcywh, normalised),
    "size" (h, w) and "labels"; text_embed (classes, output_dim); logit_scale: the CLIP parameter (log of the scale; a tensor on the
    features' device keeps the call free of host -> device copies, so that it can be captured into a HIP graph).
    Returns (list of prompts, list of logits), split per image as the reference stores them into the targets."""
    dt, dev = clip_features.dtype, clip_features.device
    coord = torch.cat([t["boxes"] for t in targets]).to(dt)
    if len(coord):
        scale = torch.cat([t["size"].flip(0).repeat(2)[None].expand(len(t["boxes"]), -1) for t in targets]).to(dt)
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
