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
import contextlib
import copy
import math

import torch
import torch.nn.functional as F
from torch import nn

from ..functions.linear import (wgrad_boundary, WgradGroup, Lin256Function, Lin256NarrowFunction, LinearBf16CachedFunction, StackedValueProjFunction, VersionCache, pack_linear256,
                                pack_linear256_padded)


def inverse_sigmoid(x, eps=1e-3):
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


class MLP(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

        self._packs = VersionCache()

    def forward(self, x):
        if x.is_cuda and x.dtype == torch.bfloat16:
            return self._forward_bf16(x)
        for i, layer in enumerate(self.layers):
            y = F.linear(x, layer.weight.to(x.dtype), layer.bias.to(x.dtype))
            x = F.relu(y) if i < self.num_layers - 1 else y
        return x

    def _forward_bf16(self, x):
        """bf16 activations, fp32 master parameters: the bf16 / packed forms of the parameters are kept across calls; layers with 256
        inputs run on csrc/lin256_mfma.hip (ReLU in the epilogue; a layer with fewer than 64 outputs padded to 64), the others on the
        library's bf16 GEMM; weight and bias gradients on the weight-gradient kernel, straight into the parameters' dtype"""
        ps = [p for layer in self.layers for p in (layer.weight, layer.bias)]

        def build():
            forms = []
            for layer in self.layers:
                w, b = layer.weight, layer.bias
                if w.shape[1] == 256 and w.shape[0] % 64 == 0:
                    forms.append(("lin256", pack_linear256([w], [b])))
                elif w.shape[1] == 256 and w.shape[0] < 64:
                    forms.append(("lin256pad", pack_linear256_padded(w, b)))
                else:
                    forms.append(("gemm", (w.detach().to(torch.bfloat16).contiguous(), b.detach().to(torch.bfloat16).contiguous())))
            return forms
        forms = self._packs.get(ps, build)
        # the 256-wide layers' weight gradients in one launch (functions/linear.py: WgradGroup) unless a caller's group is active
        wb = [(layer.weight, layer.bias) for layer in self.layers]
        lin = [i for i, (kind, _) in enumerate(forms) if kind == "lin256"]
        group = None
        if len(lin) > 1 and WgradGroup.active() is None and WgradGroup.enabled and torch.is_grad_enabled() and wb[lin[0]][0].requires_grad:
            group = WgradGroup()
            al = wgrad_boundary(group, *[t for i in lin for t in wb[i]])
            for k, i in enumerate(lin):
                wb[i] = (al[2 * k], al[2 * k + 1])
        with (group if group is not None else contextlib.nullcontext()):
            for i, (layer, (kind, f)) in enumerate(zip(self.layers, forms)):
                relu = i < self.num_layers - 1
                w, b = wb[i]
                if kind == "lin256":
                    x = Lin256Function.apply(x, f, None, relu, w, b)
                elif kind == "lin256pad" and w.shape[0] <= 8 and not relu:
                    x = Lin256NarrowFunction.apply(x, f, w, b)
                elif kind == "lin256pad":
                    x = Lin256Function.apply(x, f, None, relu, w, b)[..., :w.shape[0]]
                else:
                    x = LinearBf16CachedFunction.apply(x, f[0], f[1], None, w, b)
                    x = F.relu(x) if relu else x
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


class BoxRefineFunction(torch.autograd.Function):
    """``(delta + inverse_sigmoid(ref)).sigmoid()`` (deformable_transformer.py:779-804, richsem.py:705-715) as one launch each way
    (``msda_box_refine_forward`` / ``msda_box_refine_backward_ref``): delta bf16 or fp32 on the GPU, ref fp32 -> fp32 boxes; gradients for
    delta and -- where it carries one: the heads' boxes of decoder layers 1..5, whose reference is the previous layer's un-detached box --
    for ref (inverse_sigmoid's clamps differentiated as torch does)"""

    @staticmethod
    def forward(ctx, delta, ref, eps):
        from .. import _lib
        d, r = delta.contiguous(), ref.detach().float().contiguous()
        y = torch.empty(d.shape, dtype=torch.float32, device=d.device)
        with _lib.on_device(d.device):
            _lib.check(_lib.load().msda_box_refine_forward(d.data_ptr(), int(d.dtype == torch.bfloat16), r.data_ptr(), float(eps), d.numel(),
                                                          y.data_ptr(), _lib.raw_stream(d.device)))
        ctx.save_for_backward(y, r)
        ctx.dt, ctx.eps, ctx.ref_dt = d.dtype, float(eps), ref.dtype
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        from .. import _lib
        y, r = ctx.saved_tensors
        gy = gy.float().contiguous()
        gd = torch.empty(y.shape, dtype=ctx.dt, device=y.device)
        gr = torch.empty(y.shape, dtype=torch.float32, device=y.device) if ctx.needs_input_grad[1] else None
        with _lib.on_device(y.device):
            _lib.check(_lib.load().msda_box_refine_backward_ref(gy.data_ptr(), y.data_ptr(), y.numel(), gd.data_ptr(), int(ctx.dt == torch.bfloat16),
                                                               r.data_ptr(), ctx.eps, gr.data_ptr() if gr is not None else None,
                                                               _lib.raw_stream(y.device)))
        return gd, (gr.to(ctx.ref_dt) if gr is not None else None), None


def refine_boxes(delta, ref, eps=1e-3):
    """``(delta + inverse_sigmoid(ref)).sigmoid()``: on the library's kernels (float32 result) for bf16 / fp32 operands of equal shape on
    the GPU, the reference's op sequence otherwise"""
    if delta.is_cuda and delta.dtype in (torch.bfloat16, torch.float32) and ref.dtype == torch.float32 and delta.shape == ref.shape:
        return BoxRefineFunction.apply(delta, ref, eps)
    return (delta.to(ref.dtype) + inverse_sigmoid(ref, eps)).sigmoid()


def sine_embed_bf16(boxes, pe_dim=128):
    """:func:`gen_sineembed_for_position` of detached boxes (..., 2 | 4) f32 on the GPU -> (..., pe_dim * 2 | 4) bf16 in one launch
    (``msda_sine_embed_bf16``); no gradient (the decoder applies it to detached boxes)"""
    from .. import _lib
    b = boxes.detach()
    dims = b.shape[-1]
    # rows at a uniform stride are read in place (the level-0 slice of the (bs, nq, L, 4) boxes: stride L * 4); anything else is copied
    uniform = b.dtype == torch.float32 and b.dim() >= 2 and b.stride(-1) == 1 and all(
        b.stride(i) == b.stride(i + 1) * b.shape[i + 1] for i in range(b.dim() - 2))
    if not uniform:
        b = b.float().contiguous()
    ld, rows = (b.stride(-2), b.numel() // dims) if b.dim() >= 2 else (dims, 1)
    out = torch.empty(b.shape[:-1] + (dims * pe_dim,), dtype=torch.bfloat16, device=b.device)
    with _lib.on_device(b.device):
        _lib.check(_lib.load().msda_sine_embed_bf16(b.data_ptr(), ld, rows, dims, pe_dim, 10000.0, out.data_ptr(),
                                                   _lib.raw_stream(b.device)))
    return out


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

    def _forward_fast(self, tgt, memory, tgt_mask, memory_key_padding_mask, refpoints_unsigmoid, level_start_index, spatial_shapes,
                      valid_ratios):
        """bf16: the same computation on BATCH-first tensors throughout (the reference transposes every layer's cross-attention in
        and out, :1017-1020; its results are handed out batch-first anyway, :818-821), the memory projected once for all layers"""
        values = self._project_memory(memory, memory_key_padding_mask)
        x = tgt.transpose(0, 1).contiguous()                                           # (bs, nq, C)
        reference_points = refpoints_unsigmoid.transpose(0, 1).sigmoid()               # (bs, nq, 4)
        ref_points = [reference_points]
        intermediate = []
        vr = torch.cat([valid_ratios, valid_ratios], -1)[:, None] if reference_points.shape[-1] == 4 else valid_ratios[:, None]   # (bs, 1, L, 4)
        nw, nb = self.norm.weight.to(torch.bfloat16), self.norm.bias.to(torch.bfloat16)
        for layer_id, layer in enumerate(self.layers):
            reference_points_input = (reference_points[:, :, None] * vr.to(reference_points.dtype)).contiguous()                    # (bs, nq, L, 4)
            if reference_points_input.requires_grad:      # (not in the shipped flow: the boxes are detached between layers, :779-804)
                query_sine_embed = gen_sineembed_for_position(reference_points_input[:, :, 0, :], self.d_model // 2).to(torch.bfloat16)
            else:
                query_sine_embed = sine_embed_bf16(reference_points_input[:, :, 0, :], self.d_model // 2)
            query_pos = self.ref_point_head(query_sine_embed)
            x = layer.forward_batch_first(x, query_pos, reference_points_input.float(), values[layer_id], level_start_index, spatial_shapes,
                                          tgt_mask)
            if self.bbox_embed is not None:
                new_reference_points = refine_boxes(self.bbox_embed[layer_id](x), reference_points)
                reference_points = new_reference_points.detach()
                ref_points.append(new_reference_points)
            intermediate.append(F.layer_norm(x, (x.shape[-1],), nw, nb, self.norm.eps))
        return [intermediate, ref_points]

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None, memory_key_padding_mask=None, pos=None,
                refpoints_unsigmoid=None, level_start_index=None, spatial_shapes=None, valid_ratios=None):
        """tgt (nq, bs, C); memory (S, bs, C); refpoints_unsigmoid (nq, bs, 4); valid_ratios (bs, L, 2).  Returns the reference's
        ``[[norm(layer output) (bs, nq, C) per layer], [reference boxes (bs, nq, 4): initial + one per layer]]``."""
        # (the library-kernel path takes what the shipped configuration passes: a square boolean self-attention mask or none, no
        # cross-attention mask, no per-token padding of the queries (the reference ignores tgt_key_padding_mask too, :974-978), no memory
        # positions; anything else -- a float / additive mask as nn.MultiheadAttention accepts -- takes the reference's op sequence)
        mask_ok = tgt_mask is None or (tgt_mask.dtype == torch.bool and tgt_mask.dim() == 2 and tgt_mask.shape[0] == tgt_mask.shape[1])
        fast = (all(layer._fast(tgt) for layer in self.layers) and memory.dtype == torch.bfloat16 and self.norm is not None and mask_ok
                and memory_mask is None and tgt_key_padding_mask is None and pos is None)
        if fast:
            return self._forward_fast(tgt, memory, tgt_mask, memory_key_padding_mask, refpoints_unsigmoid, level_start_index,
                                      spatial_shapes, valid_ratios)
        output = tgt
        intermediate = []
        reference_points = refpoints_unsigmoid.sigmoid()
        ref_points = [reference_points]
        vr = torch.cat([valid_ratios, valid_ratios], -1)[None, :] if reference_points.shape[-1] == 4 else valid_ratios[None, :]
        for layer_id, layer in enumerate(self.layers):
            reference_points_input = reference_points[:, :, None] * vr.to(reference_points.dtype)              # (nq, bs, L, 4) :726-731
            query_sine_embed = gen_sineembed_for_position(reference_points_input[:, :, 0, :], self.d_model // 2)           # :734
            query_pos = self.ref_point_head(query_sine_embed.to(output.dtype))                                 # :741 (query_scale removed)
            output = layer(tgt=output, tgt_query_pos=query_pos, tgt_query_sine_embed=query_sine_embed,
                           tgt_key_padding_mask=tgt_key_padding_mask, tgt_reference_points=reference_points_input, memory=memory,
                           memory_key_padding_mask=memory_key_padding_mask, memory_level_start_index=level_start_index,
                           memory_spatial_shapes=spatial_shapes, memory_pos=pos, self_attn_mask=tgt_mask, cross_attn_mask=memory_mask)
            if self.bbox_embed is not None:                                                                    # :779-804
                reference_before_sigmoid = inverse_sigmoid(reference_points)
                delta_unsig = self.bbox_embed[layer_id](output).to(reference_points.dtype)
                new_reference_points = (delta_unsig + reference_before_sigmoid).sigmoid()
                reference_points = new_reference_points.detach()
                ref_points.append(new_reference_points)
            intermediate.append(F.layer_norm(output, (output.shape[-1],), self.norm.weight.to(output.dtype), self.norm.bias.to(output.dtype),
                                             self.norm.eps) if self.norm is not None else output)
        return [[o.transpose(0, 1) for o in intermediate], [r.transpose(0, 1) for r in ref_points]]
