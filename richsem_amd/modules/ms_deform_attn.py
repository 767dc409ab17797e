"""``MSDeformAttn`` -- host-side mirror of the reference module
(reference models/richsem/ops/modules/ms_deform_attn.py:30-115), built on the gfx950 operator.

Kept identical to the reference so released checkpoints load and callers run unchanged
(deformable_transformer.py:840,902,925 construct it; :870,:1017 call it):
  * constructor ``MSDeformAttn(d_model=256, n_levels=4, n_heads=8, n_points=4)``, ``im2col_step = 64``
  * parameter names ``sampling_offsets``, ``attention_weights``, ``value_proj``, ``output_proj``
  * initialisation (``_reset_parameters``, reference :62-76): zero offset weights, offset bias = the
    n_heads compass directions scaled to max-norm 1 times (point index + 1); zero attention
    weights; Xavier-uniform value/output projections with zero bias
  * ``forward(query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
    input_padding_mask=None)`` with 2-d (x, y) or 4-d (x, y, w, h) reference points (reference :78-115)
"""
import math
import warnings

import torch
import torch.nn.functional as F
from torch import nn

from ..functions import MaskRows, MSDeformAttnFunction, MSDeformAttnFusedFunction
from ..functions.linear import wgrad_boundary, WgradGroup, Lin256Function, LinearBf16CachedFunction, VersionCache, pack_linear256


def _is_power_of_2(n):
    if not isinstance(n, int) or n < 0:
        raise ValueError(f"invalid input for _is_power_of_2: {n} (type: {type(n)})")
    return n != 0 and (n & (n - 1)) == 0


class MSDeformAttn(nn.Module):
    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError(f"d_model must be divisible by n_heads, but got {d_model} and {n_heads}")
        if not _is_power_of_2(d_model // n_heads):
            warnings.warn("MSDeformAttn: a per-head dimension that is a power of 2 (32 in RichSem) selects the "
                          "fastest gfx950 kernels; other sizes run on the generic kernels.")
        self.im2col_step = 64
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        # True (default): one GEMM for offsets + attention logits, softmax / location arithmetic / padding mask in the library's
        # own kernels (functions/fused.py).  False: the reference's op-by-op sequence.  Same parameters, same results.
        self.fused = True
        self._bf16_ver = self._bf16_cache = None
        self._packs = VersionCache()      # lin256 forms of the four projections (d_model = 256)

        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        H, L, P = self.n_heads, self.n_levels, self.n_points
        angle = torch.arange(H, dtype=torch.float32) * (2.0 * math.pi / H)
        direction = torch.stack([angle.cos(), angle.sin()], dim=-1)
        direction = direction / direction.abs().max(dim=-1, keepdim=True)[0]          # max-norm 1
        steps = torch.arange(1, P + 1, dtype=torch.float32).view(1, 1, P, 1)
        bias = direction.view(H, 1, 1, 2).expand(H, L, P, 2) * steps
        with torch.no_grad():
            self.sampling_offsets.weight.zero_()
            self.sampling_offsets.bias = nn.Parameter(bias.reshape(-1).clone())
            self.attention_weights.weight.zero_()
            self.attention_weights.bias.zero_()
            nn.init.xavier_uniform_(self.value_proj.weight)
            self.value_proj.bias.zero_()
            nn.init.xavier_uniform_(self.output_proj.weight)
            self.output_proj.bias.zero_()
        self.invalidate_bf16_cache()

    def invalidate_bf16_cache(self):
        """Drop the cached bf16 forms of the parameters.  They are refreshed by themselves after every in-place update autograd's
        version counter sees (optimizer steps, ``load_state_dict``, ``copy_``); writes THROUGH ``param.data`` (``constant_(w.data, ...)``,
        some third-party optimizers, manual EMA / weight surgery) bypass that counter: call this after them."""
        self._bf16_ver = None
        self._bf16_cache = None
        if hasattr(self, "_packs"):
            self._packs.clear()

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self.invalidate_bf16_cache()

    def _bf16_params(self):
        """bf16 forms of the projections' parameters (offsets and logits stacked into one 256 -> 384 projection), rebuilt when any of
        them has been modified in place (optimizer step, load_state_dict; see :meth:`invalidate_bf16_cache` for ``.data`` writes)"""
        ps = (self.value_proj.weight, self.value_proj.bias, self.sampling_offsets.weight, self.sampling_offsets.bias,
              self.attention_weights.weight, self.attention_weights.bias, self.output_proj.weight, self.output_proj.bias)
        ver = tuple((p.data_ptr(), p._version) for p in ps)
        if getattr(self, "_bf16_ver", None) != ver:
            with torch.no_grad():
                b16 = lambda t: t.detach().to(torch.bfloat16).contiguous()
                self._bf16_cache = {
                    "wv": b16(ps[0]), "bv": b16(ps[1]), "wq": b16(torch.cat((ps[2], ps[4]), 0)), "bq": b16(torch.cat((ps[3], ps[5]), 0)),
                    "wo": b16(ps[6]), "bo": b16(ps[7])}
            self._bf16_ver = ver
        return self._bf16_cache

    def _lin256_packs(self):
        ps = (self.value_proj.weight, self.value_proj.bias, self.sampling_offsets.weight, self.sampling_offsets.bias,
              self.attention_weights.weight, self.attention_weights.bias, self.output_proj.weight, self.output_proj.bias)
        return self._packs.get(ps, lambda: {"v": pack_linear256([ps[0]], [ps[1]]), "q": pack_linear256([ps[2], ps[4]], [ps[3], ps[5]]),
                                            "o": pack_linear256([ps[6]], [ps[7]])})

    def project_value(self, input_flatten, input_padding_mask=None, params=None):
        """bf16, d_model = 256: ``value_proj(input_flatten)`` with the rows of padded pixels zeroed (reference :94-96), (N, S, C);
        ``params``: stand-ins for (value_proj.weight, value_proj.bias) to route the gradients to"""
        pk = self._lin256_packs()
        mask = input_padding_mask.contiguous() if input_padding_mask is not None else None
        vw, vb = params if params is not None else (self.value_proj.weight, self.value_proj.bias)
        return Lin256Function.apply(input_flatten.to(torch.bfloat16), pk["v"], mask, False, vw, vb)

    def forward_from_value(self, query, reference_points, value, input_spatial_shapes, input_level_start_index, params=None):
        """bf16, d_model = 256: the module's forward behind the value projection (reference :97-114) -- for callers that project the
        memory for several layers at once (richsem_amd/modules/decoder.py).  ``params``: stand-ins for (sampling_offsets.weight,
        attention_weights.weight, sampling_offsets.bias, attention_weights.bias, output_proj.weight, output_proj.bias) to route the
        gradients to (a layer's WgradBoundary aliases: functions/linear.py)"""
        N, S = value.shape[0], value.shape[1]
        H, L, P = self.n_heads, self.n_levels, self.n_points
        pk = self._lin256_packs()
        if params is None:
            params = (self.sampling_offsets.weight, self.attention_weights.weight, self.sampling_offsets.bias, self.attention_weights.bias,
                      self.output_proj.weight, self.output_proj.bias)
        qproj = Lin256Function.apply(query.to(torch.bfloat16), pk["q"], None, False, *params[:4])
        out = MSDeformAttnFusedFunction.apply(value.reshape(N, S, H, self.d_model // H), input_spatial_shapes, input_level_start_index,
                                              qproj, reference_points.float(), H, L, P, self.im2col_step)
        return Lin256Function.apply(out, pk["o"], None, False, *params[4:])

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None):
        """query (N, Lq, C); reference_points (N, Lq, L, 2|4) in [0,1] incl. padding; input_flatten (N, S, C);
        input_spatial_shapes (L, 2) = (H_l, W_l); input_level_start_index (L,); input_padding_mask (N, S) bool,
        True on padding.  Returns (N, Lq, C)."""
        N, Lq, _ = query.shape
        S = input_flatten.shape[1]
        H, L, P = self.n_heads, self.n_levels, self.n_points
        if input_flatten.is_cuda:      # the reference's assert (ms_deform_attn.py:99) on the cached host mirror: no stream synchronisation per call
            from ..MultiScaleDeformableAttention import _host_mirror
            hs = _host_mirror(input_spatial_shapes, input_level_start_index)[0]
            assert int((hs[:, 0] * hs[:, 1]).sum()) == S
        else:
            assert (input_spatial_shapes[:, 0] * input_spatial_shapes[:, 1]).sum() == S

        if (self.fused and query.is_cuda and query.dtype in (torch.float32, torch.float64, torch.bfloat16) and L * P <= 64
                and L <= 16):
            if reference_points.shape[-1] not in (2, 4):
                raise ValueError(
                    f"Last dim of reference_points must be 2 or 4, but get {reference_points.shape[-1]} instead.")
            # bfloat16 activations (new capability: the reference has no half path): the four projections run as bf16 GEMMs on the
            # module's (fp32) parameters cast per call, value / output travel as bf16 through the operator's bf16 entry points,
            # locations and attention weights are formed and kept in fp32
            dt = query.dtype
            if dt == torch.bfloat16 and self.d_model == 256 and (H * L * P * 3) % 64 == 0:
                # bf16, d_model = 256 (RichSem): the four projections on the library's own MFMA kernel (csrc/lin256_mfma.hip: K = 256) --
                # value_proj with the padding mask in its epilogue, offsets + logits as ONE 256 -> 384 projection, output_proj; their
                # input gradients on the same kernel where the layer is 256 -> 256, the weight gradients on the weight-gradient kernel
                # (the three weight gradients in one launch -- functions/linear.py: WgradGroup -- unless a caller's group is active)
                group = None
                if WgradGroup.active() is None and WgradGroup.enabled and torch.is_grad_enabled() and self.value_proj.weight.requires_grad:
                    group = WgradGroup()
                    al = wgrad_boundary(group, self.value_proj.weight, self.value_proj.bias, self.sampling_offsets.weight,
                                             self.attention_weights.weight, self.sampling_offsets.bias, self.attention_weights.bias,
                                             self.output_proj.weight, self.output_proj.bias)
                    with group:
                        return self.forward_from_value(query, reference_points, self.project_value(input_flatten, input_padding_mask, al[:2]),
                                                       input_spatial_shapes, input_level_start_index, params=al[2:])
                return self.forward_from_value(query, reference_points, self.project_value(input_flatten, input_padding_mask),
                                               input_spatial_shapes, input_level_start_index)
            if dt == torch.bfloat16:
                # bf16, other widths: library GEMMs for the forward and the input gradient, the library's own MFMA kernel for the weight
                # gradient (its contraction runs over the tokens); the bf16 casts of the parameters and the stacked offsets / logits
                # projection are kept across calls (refreshed when a parameter changes)
                c = self._bf16_params()
                value = LinearBf16CachedFunction.apply(input_flatten.to(dt), c["wv"], c["bv"], None, self.value_proj.weight,
                                                       self.value_proj.bias)
                if input_padding_mask is not None:
                    value = MaskRows.apply(value, input_padding_mask)
                qproj = LinearBf16CachedFunction.apply(query, c["wq"], c["bq"], self.sampling_offsets.weight.shape[0],
                                                       self.sampling_offsets.weight, self.attention_weights.weight,
                                                       self.sampling_offsets.bias, self.attention_weights.bias)
                out = MSDeformAttnFusedFunction.apply(value.view(N, S, H, self.d_model // H), input_spatial_shapes,
                                                      input_level_start_index, qproj, reference_points.float(), H, L, P, self.im2col_step)
                return LinearBf16CachedFunction.apply(out, c["wo"], c["bo"], None, self.output_proj.weight, self.output_proj.bias)
            value = F.linear(input_flatten.to(dt), self.value_proj.weight, self.value_proj.bias)
            if input_padding_mask is not None:
                value = MaskRows.apply(value, input_padding_mask)
            # offsets and attention logits from ONE projection (the two weight matrices stacked: 256 -> 384 for RichSem)
            weight = torch.cat((self.sampling_offsets.weight, self.attention_weights.weight), 0)
            bias = torch.cat((self.sampling_offsets.bias, self.attention_weights.bias), 0)
            qproj = F.linear(query, weight, bias)
            out = MSDeformAttnFusedFunction.apply(value.view(N, S, H, self.d_model // H), input_spatial_shapes,
                                                  input_level_start_index, qproj, reference_points.to(dt), H, L, P,
                                                  self.im2col_step)
            return F.linear(out, self.output_proj.weight, self.output_proj.bias)

        value = self.value_proj(input_flatten)
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask[..., None], float(0))
        value = value.view(N, S, H, self.d_model // H)

        offsets = self.sampling_offsets(query).view(N, Lq, H, L, P, 2)
        weights = F.softmax(self.attention_weights(query).view(N, Lq, H, L * P), -1).view(N, Lq, H, L, P)

        ref = reference_points[:, :, None, :, None, :]                                   # (N, Lq, 1, L, 1, 2|4)
        if reference_points.shape[-1] == 2:
            wh = torch.stack([input_spatial_shapes[..., 1], input_spatial_shapes[..., 0]], -1)   # (L, 2) = (W, H)
            locations = ref + offsets / wh[None, None, None, :, None, :]
        elif reference_points.shape[-1] == 4:
            locations = ref[..., :2] + offsets / P * ref[..., 2:] * 0.5
        else:
            raise ValueError(
                f"Last dim of reference_points must be 2 or 4, but get {reference_points.shape[-1]} instead.")

        out = MSDeformAttnFunction.apply(value, input_spatial_shapes, input_level_start_index, locations, weights,
                                         self.im2col_step)
        return self.output_proj(out)
