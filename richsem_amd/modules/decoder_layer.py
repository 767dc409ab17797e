"""Mirror of the reference's ``DeformableTransformerDecoderLayer`` (models/richsem/deformable_transformer.py:883-1066) in its
shipped configuration -- ``module_seq = ['sa', 'ca', 'ffn']``, ``decoder_sa_type = 'sa'``, no key-aware projection, no box
attention:

    tgt = norm2(tgt + dropout2(self_attn(q = k = tgt + query_pos, v = tgt, attn_mask)))           nn.MultiheadAttention
    tgt = norm1(tgt + dropout1(cross_attn(tgt + query_pos, reference boxes, memory, shapes, ...)))  MSDeformAttn, 4-d references
    tgt = norm3(tgt + dropout4(linear2(dropout3(activation(linear1(tgt))))))                        feed-forward block

Sequence-first tensors (nq, bs, d_model) like the reference.  Same parameter names (``cross_attn.*``, ``self_attn.in_proj_weight``
/ ``in_proj_bias`` / ``out_proj.*``, ``norm1..3``, ``linear1/2``), so a reference checkpoint loads unchanged.

Two paths:
  * float32 / float64 activations: the reference's op sequence with ``cross_attn`` on the HIP kernels (what the reference-generated
    fixtures of tests/golden/layer_decoder_f64.npz pin);
  * bfloat16 activations, d_model = 256, relu, no active dropout (new capability: the reference has no half path) -- every product on
    the library's kernels: q/k and v projections, the attention's output projection, the cross-attention's projections and the
    feed-forward block's first product on ``csrc/lin256_mfma.hip``; the masked self-attention as one kernel each way
    (``csrc/attn_mfma.hip``); residual + LayerNorm fused (``AddLayerNormFunction``); the cross-attention's value can be handed in by
    the decoder, which projects the memory for all its layers at once (modules/decoder.py).
"""
import contextlib

import torch
import torch.nn.functional as F
from torch import nn

from ..functions.attention import masked_self_attention
from ..functions.ffn import AddLayerNormFunction, FFNSmallFunction
from ..functions.linear import Lin256Function, VersionCache, wgrad_boundary, WgradGroup, lin256_pack, pack_linear256
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
        self.n_heads = n_heads
        self.fused = True                 # False: bfloat16 input takes the op-by-op sequence too (parameters cast per call)
        self.group_wgrad = True           # the fast path's seven weight gradients in one launch (functions/linear.py: WgradGroup)
        self._packs = VersionCache()

    @staticmethod
    def with_pos_embed(tensor, pos):
        return tensor if pos is None else tensor + pos

    # ---- the library-kernel path (bfloat16) ---------------------------------------------------------------------------------------
    def _fast(self, tgt):
        drop = self.training and max(self.dropout1.p, self.dropout2.p, self.dropout3.p, self.dropout4.p, self.self_attn.dropout) > 0
        return (self.fused and tgt.is_cuda and tgt.dtype == torch.bfloat16 and tgt.shape[-1] == 256 and self.activation == "relu"
                and not drop and self.linear1.out_features % 64 == 0 and self.self_attn.in_proj_weight is not None)

    def invalidate_bf16_cache(self):
        """after writes through ``param.data`` (see functions/linear.py: VersionCache)"""
        self._packs.clear()
        self.cross_attn.invalidate_bf16_cache()

    def _lin_packs(self):
        a = self.self_attn
        ps = (a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias, self.linear1.weight, self.linear1.bias,
              self.linear2.weight, self.linear2.bias)

        def build():
            w, b = a.in_proj_weight, a.in_proj_bias
            return {"qk": pack_linear256([w[:512]], [b[:512]]), "v": pack_linear256([w[512:]], [b[512:]]),
                    "o": pack_linear256([a.out_proj.weight], [a.out_proj.bias]),
                    "w1": pack_linear256([self.linear1.weight], [self.linear1.bias]),
                    "w2_16": self.linear2.weight.detach().to(torch.bfloat16).contiguous(),
                    "w2t": lin256_pack(self.linear2.weight.detach().to(torch.bfloat16).t().contiguous())}
        return self._packs.get(ps, build)

    def forward_batch_first(self, x, query_pos, reference_points, value, lsi, shapes, attn_mask):
        """the bf16 library-kernel path on BATCH-first tensors (no transposes between its blocks): x, query_pos (bs, nq, 256) bf16,
        reference_points (bs, nq, L, 4) float32, value = the cross-attention's projected memory (bs, S, 256) bf16 -> (bs, nq, 256)"""
        pk = self._lin_packs()
        a, ca = self.self_attn, self.cross_attn
        w, b, ow, ob = a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias
        l1w, l1b, l2w = self.linear1.weight, self.linear1.bias, self.linear2.weight
        cross = (ca.sampling_offsets.weight, ca.attention_weights.weight, ca.sampling_offsets.bias, ca.attention_weights.bias, ca.output_proj.weight,
                 ca.output_proj.bias)
        # the layer's seven weight gradients in ONE launch (functions/linear.py: WgradGroup): the parameters go through a boundary whose
        # backward runs behind every function below; each alias feeds exactly one of them (in_proj's two slices are one alias each)
        group = WgradGroup() if (self.group_wgrad and WgradGroup.enabled and torch.is_grad_enabled() and w.requires_grad) else None
        if group is not None:
            wqk, wv, bqk, bv, ow, ob, l1w, l1b, l2w, *cross = wgrad_boundary(group, w[:512], w[512:], b[:512], b[512:], ow, ob, l1w, l1b, l2w, *cross)
        else:
            wqk, wv, bqk, bv = w[:512], w[512:], b[:512], b[512:]
        with (group if group is not None else contextlib.nullcontext()):
            # self-attention (:974-978)
            q_in = x if query_pos is None else x + query_pos
            qk = Lin256Function.apply(q_in, pk["qk"], None, False, wqk, bqk)
            v = Lin256Function.apply(x, pk["v"], None, False, wv, bv)
            att = masked_self_attention(qk, v, attn_mask, self.n_heads, batch_first=True)
            x2 = Lin256Function.apply(att, pk["o"], None, False, ow, ob)
            x = AddLayerNormFunction.apply(x, x2, self.norm2.weight, self.norm2.bias, self.norm2.eps)
            # cross-attention (:1017-1022)
            q_in = x if query_pos is None else x + query_pos
            x2 = self.cross_attn.forward_from_value(q_in, reference_points, value, shapes, lsi, params=tuple(cross))
            x = AddLayerNormFunction.apply(x, x2, self.norm1.weight, self.norm1.bias, self.norm1.eps)
            # feed-forward block (:940-944)
            return FFNSmallFunction.apply(x, pk["w1"], pk["w2_16"], pk["w2t"], self.norm3.eps, l1w, l1b, l2w, self.linear2.bias, self.norm3.weight,
                                          self.norm3.bias)

    def _forward_fast(self, tgt, query_pos, reference_points, memory, memory_mask, lsi, shapes, attn_mask, value):
        """sequence-first in and out (the reference's layout) around :meth:`forward_batch_first`"""
        if value is None:
            value = self.cross_attn.project_value(memory.transpose(0, 1), memory_mask)
        qpos = query_pos.to(torch.bfloat16).transpose(0, 1).contiguous() if query_pos is not None else None
        out = self.forward_batch_first(tgt.transpose(0, 1).contiguous(), qpos, reference_points.transpose(0, 1).contiguous(), value, lsi,
                                       shapes, attn_mask)
        return out.transpose(0, 1).contiguous()

    # ---- the reference's op sequence --------------------------------------------------------------------------------------------
    def forward_ffn(self, tgt):
        act = {"relu": F.relu, "gelu": F.gelu}[self.activation]
        dt = tgt.dtype
        h = self.dropout3(act(F.linear(tgt, self.linear1.weight.to(dt), self.linear1.bias.to(dt))))
        tgt2 = F.linear(h, self.linear2.weight.to(dt), self.linear2.bias.to(dt))
        return F.layer_norm(tgt + self.dropout4(tgt2), (tgt.shape[-1],), self.norm3.weight.to(dt), self.norm3.bias.to(dt), self.norm3.eps)

    def forward_sa(self, tgt, tgt_query_pos=None, self_attn_mask=None):
        q = k = self.with_pos_embed(tgt, tgt_query_pos)
        a, dt = self.self_attn, tgt.dtype
        if dt == a.in_proj_weight.dtype:
            tgt2 = a(q, k, tgt, attn_mask=self_attn_mask)[0]
        else:   # bf16 activations with fp32 master parameters, op by op
            tgt2 = F.multi_head_attention_forward(q, k, tgt, a.embed_dim, a.num_heads, a.in_proj_weight.to(dt), a.in_proj_bias.to(dt), None, None,
                                                  False, a.dropout, a.out_proj.weight.to(dt), a.out_proj.bias.to(dt), training=self.training,
                                                  attn_mask=self_attn_mask, need_weights=False)[0]
        dt = tgt.dtype
        return F.layer_norm(tgt + self.dropout2(tgt2), (tgt.shape[-1],), self.norm2.weight.to(dt), self.norm2.bias.to(dt), self.norm2.eps)

    def forward_ca(self, tgt, tgt_query_pos, tgt_reference_points, memory, memory_key_padding_mask, memory_level_start_index,
                   memory_spatial_shapes):
        tgt2 = self.cross_attn(self.with_pos_embed(tgt, tgt_query_pos).transpose(0, 1), tgt_reference_points.transpose(0, 1).contiguous(),
                               memory.transpose(0, 1), memory_spatial_shapes, memory_level_start_index,
                               memory_key_padding_mask).transpose(0, 1)
        dt = tgt.dtype
        return F.layer_norm(tgt + self.dropout1(tgt2), (tgt.shape[-1],), self.norm1.weight.to(dt), self.norm1.bias.to(dt), self.norm1.eps)

    def forward(self, tgt, tgt_query_pos=None, tgt_query_sine_embed=None, tgt_key_padding_mask=None, tgt_reference_points=None,
                memory=None, memory_key_padding_mask=None, memory_level_start_index=None, memory_spatial_shapes=None,
                memory_pos=None, self_attn_mask=None, cross_attn_mask=None, value=None):
        """the reference's signature (:1026-1043) plus ``value``: the cross-attention's projected memory when the caller has it"""
        mask_ok = self_attn_mask is None or (self_attn_mask.dtype == torch.bool and self_attn_mask.dim() == 2
                                             and self_attn_mask.shape[0] == self_attn_mask.shape[1])
        if self._fast(tgt) and mask_ok:      # (a float / additive mask takes the reference's op sequence: nn.MultiheadAttention accepts it)
            return self._forward_fast(tgt, tgt_query_pos, tgt_reference_points, memory, memory_key_padding_mask, memory_level_start_index,
                                      memory_spatial_shapes, self_attn_mask, value)
        tgt = self.forward_sa(tgt, tgt_query_pos, self_attn_mask)
        tgt = self.forward_ca(tgt, tgt_query_pos, tgt_reference_points, memory, memory_key_padding_mask, memory_level_start_index,
                              memory_spatial_shapes)
        return self.forward_ffn(tgt)
