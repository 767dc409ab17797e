"""Mirror of the reference's ``TransformerDecoder`` (models/richsem/deformable_transformer.py:620-823) in the shipped configuration
(``deformable_decoder=True``, ``query_dim=4``, ``rm_dec_query_scale=True``, ``return_intermediate=True``, no query perturber, no layer
dropout, no per-layer query selection) and of the helpers it calls: ``gen_sineembed_for_position`` (models/richsem/utils.py:142-168),
``MLP`` (:110-122), ``inverse_sigmoid`` (util/misc.py:605-609).

    per layer:  reference boxes x valid ratios -> sine embedding of the level-0 box -> ``ref_point_head`` MLP -> query_pos
                layer(tgt, query_pos, reference boxes, memory, ...)                                   (modules/decoder_layer.py)
                ``bbox_embed[layer]`` refines the (detached) boxes for the next layer                    (:779-804)
                ``norm`` of the layer output is collected                                               (:810)

Same parameter names as the reference (``layers.{i}.*``, ``norm.*``, ``ref_point_head.layers.{j}.*``, ``bbox_embed.{i}.layers.{j}.*``),
so the ``transformer.decoder.*`` entries of a reference checkpoint load unchanged.

bfloat16 memory (new capability): the cross-attentions' value projections of ALL layers are ONE product -- the memory is read once,
256 -> 256 x num_layers on ``csrc/lin256_mfma.hip`` with the padding mask in its epilogue -- and the memory's gradient is one product
too; each layer takes its slice.  (The reference projects the 22 k memory tokens separately in each of its six layers.)
"""
import copy
import math

import torch
import torch.nn.functional as F
from torch import nn

from ..functions.linear import StackedValueProjFunction, VersionCache, pack_linear256


def inverse_sigmoid(x, eps=1e-3):
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


class MLP(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            y = F.linear(x, layer.weight.to(x.dtype), layer.bias.to(x.dtype))
            x = F.relu(y) if i < self.num_layers - 1 else y
        return x


def gen_sineembed_for_position(pos_tensor, pe_dim=128):
    """(nq, bs, 2 | 4) boxes -> (nq, bs, pe_dim * 2 | 4) in the order (y, x[, w, h]); sin on the even, cos on the odd channels"""
    scale = 2 * math.pi
    dim_t = torch.arange(pe_dim, dtype=torch.float32, device=pos_tensor.device)
    dim_t = 10000 ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / pe_dim)

    def emb(coord):
        p = (coord * scale)[:, :, None] / dim_t
        return torch.stack((p[:, :, 0::2].sin(), p[:, :, 1::2].cos()), dim=3).flatten(2)

    parts = [emb(pos_tensor[:, :, 1]), emb(pos_tensor[:, :, 0])]
    if pos_tensor.size(-1) == 4:
        parts += [emb(pos_tensor[:, :, 2]), emb(pos_tensor[:, :, 3])]
    elif pos_tensor.size(-1) != 2:
        raise ValueError("Unknown pos_tensor shape(-1):{}".format(pos_tensor.size(-1)))
    return torch.cat(parts, dim=2)


class TransformerDecoder(nn.Module):
    def __init__(self, decoder_layer, num_layers, norm=None, d_model=256, query_dim=4, num_feature_levels=4):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(decoder_layer) for _ in range(num_layers)])
        self.num_layers = num_layers
        self.norm = norm
        self.query_dim = query_dim
        self.num_feature_levels = num_feature_levels
        self.d_model = d_model
        self.ref_point_head = MLP(query_dim // 2 * d_model, d_model, d_model, 2)
        self.bbox_embed = None      # nn.ModuleList of MLP(d_model, d_model, 4, 3), one per layer: attached by the model (richsem.py)
        self.class_embed = None
        self._value_packs = VersionCache()

    def invalidate_bf16_cache(self):
        self._value_packs.clear()
        for layer in self.layers:
            layer.invalidate_bf16_cache()

    def _project_memory(self, memory, memory_key_padding_mask):
        """bf16: value_proj of every layer's cross-attention in one product; returns the per-layer (N, S, 256) values"""
        mods = [layer.cross_attn for layer in self.layers]
        ws, bs_ = [m.value_proj.weight for m in mods], [m.value_proj.bias for m in mods]
        pk = self._value_packs.get(tuple(ws + bs_), lambda: pack_linear256(ws, bs_))
        mem = memory.transpose(0, 1)                                     # (N, S, C): batch-first, as the module takes it
        mask = memory_key_padding_mask.contiguous() if memory_key_padding_mask is not None else None
        return StackedValueProjFunction.apply(mem.contiguous(), pk, mask, *ws, *bs_)     # one (N, S, 256) tensor per layer

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None, memory_key_padding_mask=None, pos=None,
                refpoints_unsigmoid=None, level_start_index=None, spatial_shapes=None, valid_ratios=None):
        """tgt (nq, bs, C); memory (S, bs, C); refpoints_unsigmoid (nq, bs, 4); valid_ratios (bs, L, 2).  Returns the reference's
        ``[[norm(layer output) (bs, nq, C) per layer], [reference boxes (bs, nq, 4): initial + one per layer]]``."""
        output = tgt
        intermediate = []
        reference_points = refpoints_unsigmoid.sigmoid()
        ref_points = [reference_points]
        fast = all(layer._fast(output) for layer in self.layers) and memory.dtype == torch.bfloat16
        values = self._project_memory(memory, memory_key_padding_mask) if fast else [None] * self.num_layers
        vr = torch.cat([valid_ratios, valid_ratios], -1)[None, :] if reference_points.shape[-1] == 4 else valid_ratios[None, :]
        for layer_id, layer in enumerate(self.layers):
            reference_points_input = reference_points[:, :, None] * vr.to(reference_points.dtype)              # (nq, bs, L, 4) :726-731
            query_sine_embed = gen_sineembed_for_position(reference_points_input[:, :, 0, :], self.d_model // 2)           # :734
            query_pos = self.ref_point_head(query_sine_embed.to(output.dtype))                                 # :741 (query_scale removed)
            output = layer(tgt=output, tgt_query_pos=query_pos, tgt_query_sine_embed=query_sine_embed,
                           tgt_key_padding_mask=tgt_key_padding_mask, tgt_reference_points=reference_points_input, memory=memory,
                           memory_key_padding_mask=memory_key_padding_mask, memory_level_start_index=level_start_index,
                           memory_spatial_shapes=spatial_shapes, memory_pos=pos, self_attn_mask=tgt_mask, cross_attn_mask=memory_mask,
                           value=values[layer_id])
            if self.bbox_embed is not None:                                                                    # :779-804
                reference_before_sigmoid = inverse_sigmoid(reference_points)
                delta_unsig = self.bbox_embed[layer_id](output).to(reference_points.dtype)
                new_reference_points = (delta_unsig + reference_before_sigmoid).sigmoid()
                reference_points = new_reference_points.detach()
                ref_points.append(new_reference_points)
            intermediate.append(F.layer_norm(output, (output.shape[-1],), self.norm.weight.to(output.dtype), self.norm.bias.to(output.dtype),
                                             self.norm.eps) if self.norm is not None else output)
        return [[o.transpose(0, 1) for o in intermediate], [r.transpose(0, 1) for r in ref_points]]
