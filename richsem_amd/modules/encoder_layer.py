"""Mirror of the reference's ``DeformableTransformerEncoderLayer`` (models/richsem/deformable_transformer.py:825-881) and of
the encoder's ``get_reference_points`` (:512-525), assembled from this package's kernels:

    src = norm1(src + dropout1(self_attn(src + pos, reference_points, src, spatial_shapes, level_start_index, padding_mask)))
    src = norm2(src + dropout3(linear2(dropout2(activation(linear1(src))))))

Same parameter names as the reference layer (``self_attn.*``, ``norm1``, ``linear1``, ``linear2``, ``norm2``), so a reference
checkpoint loads unchanged.  ``self_attn`` is ``richsem_amd.modules.MSDeformAttn`` (the operator on the HIP kernels, fused
module path); the feed-forward block is ONE MFMA kernel when the layer runs in bfloat16 with relu and no active dropout
(``richsem_amd/csrc/ffn_mfma.hip``), else the reference's op-by-op sequence.  The optional channel attention (``dyrelu``) and
the box-attention variant of the reference are not mirrored.
"""
import torch
import torch.nn.functional as F
from torch import nn

from ..functions.ffn import FUSED_FFN_MIN_TOKENS, AddLayerNormFunction, FusedFFNCachedFunction, FusedFFNFunction, pack_ffn
from ..functions.linear import VersionCache, linear_wgrad_supported
from .ms_deform_attn import MSDeformAttn


def get_reference_points(spatial_shapes, valid_ratios, device):
    """reference deformable_transformer.py:512-525: pixel centres of every level, normalised by the valid (unpadded) part
    of the map, then scaled to every level's valid ratio -> (N, S, L, 2)."""
    out = []
    for lvl, (H_, W_) in enumerate(spatial_shapes):
        H_, W_ = int(H_), int(W_)
        ref_y, ref_x = torch.meshgrid(torch.linspace(0.5, H_ - 0.5, H_, dtype=torch.float32, device=device),
                                      torch.linspace(0.5, W_ - 0.5, W_, dtype=torch.float32, device=device), indexing="ij")
        ref_y = ref_y.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * H_)
        ref_x = ref_x.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * W_)
        out.append(torch.stack((ref_x, ref_y), -1))
    reference_points = torch.cat(out, 1)
    return reference_points[:, :, None] * valid_ratios[:, None]


class DeformableTransformerEncoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.activation = activation
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)
        self.fused_ffn = True
        self.fused_min_tokens = FUSED_FFN_MIN_TOKENS   # below it the op sequence is faster (functions/ffn.py)
        self._ffn_pack = VersionCache()                 # packed forms of linear1 / linear2 (rebuilt when a parameter changes)

    @staticmethod
    def with_pos_embed(tensor, pos):
        return tensor if pos is None else tensor + pos

    def _ffn_fusable(self, src):
        drop = self.training and (self.dropout2.p > 0 or self.dropout3.p > 0)
        return (self.fused_ffn and src.is_cuda and src.dtype == torch.bfloat16 and self.activation == "relu" and not drop
                and src.shape[-1] == 256 and self.linear1.out_features % 32 == 0 and self.linear1.out_features <= 4096
                and src.numel() // src.shape[-1] >= self.fused_min_tokens)

    def forward_ffn(self, src):
        if self._ffn_fusable(src) and self.linear1.out_features % 128 == 0 and linear_wgrad_supported(256, self.linear1.out_features):
            pk = self._ffn_pack.get((self.linear1.weight, self.linear1.bias, self.linear2.weight),
                                    lambda: pack_ffn(self.linear1.weight, self.linear1.bias, self.linear2.weight))
            return FusedFFNCachedFunction.apply(src, pk, self.norm2.eps, self.linear1.weight, self.linear1.bias, self.linear2.weight,
                                                self.linear2.bias, self.norm2.weight, self.norm2.bias)
        if self._ffn_fusable(src):      # (hidden widths the weight-gradient kernel does not take: parameters cast / packed per call)
            return FusedFFNFunction.apply(src, self.linear1.weight.to(torch.bfloat16), self.linear1.bias.float(),
                                          self.linear2.weight.to(torch.bfloat16), self.linear2.bias.float(),
                                          self.norm2.weight.float(), self.norm2.bias.float(), self.norm2.eps)
        # op-by-op sequence (fp32, or bf16 activations with fp32 master parameters: the parameters are cast to the input's type)
        act = {"relu": F.relu, "gelu": F.gelu}[self.activation]
        dt = src.dtype
        h = self.dropout2(act(F.linear(src, self.linear1.weight.to(dt), self.linear1.bias.to(dt))))
        src2 = F.linear(h, self.linear2.weight.to(dt), self.linear2.bias.to(dt))
        return F.layer_norm(src + self.dropout3(src2), (src.shape[-1],), self.norm2.weight.to(dt), self.norm2.bias.to(dt), self.norm2.eps)

    def forward(self, src, pos, reference_points, spatial_shapes, level_start_index, key_padding_mask=None):
        # (bfloat16 activations: the attention module runs its projections as bf16 GEMMs and the operator's bf16 entry points, with
        # fp32 locations / weights -- new capability: the reference has no half path; the LayerNorms compute in fp32 internally)
        src2 = self.self_attn(self.with_pos_embed(src, pos), reference_points, src, spatial_shapes, level_start_index, key_padding_mask)
        if (self.fused_ffn and src.is_cuda and src.dtype == torch.bfloat16 and src.shape[-1] == 256
                and not (self.training and self.dropout1.p > 0)):
            # residual add + LayerNorm as one kernel each way (functions/ffn.py: AddLayerNormFunction)
            src = AddLayerNormFunction.apply(src, src2, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        else:
            src = F.layer_norm(src + self.dropout1(src2), (src.shape[-1],), self.norm1.weight.to(src.dtype), self.norm1.bias.to(src.dtype),
                               self.norm1.eps)
        return self.forward_ffn(src)
