"""Mirror of the reference's ``DeformableTransformerDecoderLayer`` (models/richsem/deformable_transformer.py:883-1066) in its
shipped configuration -- ``module_seq = ['sa', 'ca', 'ffn']``, ``decoder_sa_type = 'sa'``, no key-aware projection, no box
attention:

    tgt = norm2(tgt + dropout2(self_attn(q = k = tgt + query_pos, v = tgt, attn_mask)))           nn.MultiheadAttention
    tgt = norm1(tgt + dropout1(cross_attn(tgt + query_pos, reference boxes, memory, shapes, ...)))  MSDeformAttn, 4-d references
    tgt = norm3(tgt + dropout4(linear2(dropout3(activation(linear1(tgt))))))                        feed-forward block

Sequence-first tensors (nq, bs, d_model) like the reference.  Same parameter names (``cross_attn.*``, ``self_attn.*``,
``norm1..3``, ``linear1/2``), so a reference checkpoint loads unchanged.  ``cross_attn`` runs on the HIP kernels (decoder-shaped
calls: direct forward, level-sum + direct backward); the feed-forward block is the MFMA kernel for bfloat16 input with relu and no
active dropout, else the reference's op-by-op sequence; the self-attention is PyTorch's.
"""
import torch
import torch.nn.functional as F
from torch import nn

from ..functions.ffn import FUSED_FFN_MIN_TOKENS, FusedFFNFunction
from .ms_deform_attn import MSDeformAttn


class DeformableTransformerDecoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        self.cross_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.self_attn = nn.MultiheadAttention(d_model, n_heads, dropout=dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.activation = activation
        self.dropout3 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout4 = nn.Dropout(dropout)
        self.norm3 = nn.LayerNorm(d_model)
        self.fused_ffn = True
        self.fused_min_tokens = FUSED_FFN_MIN_TOKENS   # below it the op sequence is faster (functions/ffn.py)

    @staticmethod
    def with_pos_embed(tensor, pos):
        return tensor if pos is None else tensor + pos

    def forward_ffn(self, tgt):
        drop = self.training and (self.dropout3.p > 0 or self.dropout4.p > 0)
        if (self.fused_ffn and tgt.is_cuda and tgt.dtype == torch.bfloat16 and self.activation == "relu" and not drop
                and tgt.shape[-1] == 256 and self.linear1.out_features % 32 == 0 and self.linear1.out_features <= 4096
                and tgt.numel() // tgt.shape[-1] >= self.fused_min_tokens):
            return FusedFFNFunction.apply(tgt, self.linear1.weight.to(torch.bfloat16), self.linear1.bias.float(),
                                          self.linear2.weight.to(torch.bfloat16), self.linear2.bias.float(),
                                          self.norm3.weight.float(), self.norm3.bias.float(), self.norm3.eps)
        act = {"relu": F.relu, "gelu": F.gelu}[self.activation]
        tgt2 = self.linear2(self.dropout3(act(self.linear1(tgt))))
        return self.norm3(tgt + self.dropout4(tgt2))

    def forward_sa(self, tgt, tgt_query_pos=None, self_attn_mask=None):
        q = k = self.with_pos_embed(tgt, tgt_query_pos)
        tgt2 = self.self_attn(q, k, tgt, attn_mask=self_attn_mask)[0]
        return self.norm2(tgt + self.dropout2(tgt2))

    def forward_ca(self, tgt, tgt_query_pos, tgt_reference_points, memory, memory_key_padding_mask, memory_level_start_index,
                   memory_spatial_shapes):
        tgt2 = self.cross_attn(self.with_pos_embed(tgt, tgt_query_pos).transpose(0, 1), tgt_reference_points.transpose(0, 1).contiguous(),
                               memory.transpose(0, 1), memory_spatial_shapes, memory_level_start_index,
                               memory_key_padding_mask).transpose(0, 1)
        return self.norm1(tgt + self.dropout1(tgt2))

    def forward(self, tgt, tgt_query_pos=None, tgt_query_sine_embed=None, tgt_key_padding_mask=None, tgt_reference_points=None,
                memory=None, memory_key_padding_mask=None, memory_level_start_index=None, memory_spatial_shapes=None,
                memory_pos=None, self_attn_mask=None, cross_attn_mask=None):
        tgt = self.forward_sa(tgt, tgt_query_pos, self_attn_mask)
        tgt = self.forward_ca(tgt, tgt_query_pos, tgt_reference_points, memory, memory_key_padding_mask, memory_level_start_index,
                              memory_spatial_shapes)
        return self.forward_ffn(tgt)
