"""Autograd binding of the MSDeformAttn operator -- host-side mirror of the reference's
``MSDeformAttnFunction`` (reference models/richsem/ops/functions/ms_deform_attn_func.py:21-38):
same class name, same ``apply`` arguments, same returned gradients
``(grad_value, None, None, grad_sampling_loc, grad_attn_weight, None)``.

The reference file also carries a pure-PyTorch CPU function (``ms_deform_attn_core_pytorch``,
:41-61, "for debug and test only").  It has no counterpart here on purpose: this package has no
CPU path; the checker lives under oracle/ and is used by the tests only.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import MultiScaleDeformableAttention as MSDA


class MSDeformAttnFunction(Function):
    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                im2col_step):
        ctx.im2col_step = im2col_step
        output = MSDA.ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index,
                                             sampling_locations, attention_weights, ctx.im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                              attention_weights)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, spatial_shapes, level_start_index, sampling_locations, attention_weights = ctx.saved_tensors
        grad_value, grad_sampling_loc, grad_attn_weight = MSDA.ms_deform_attn_backward(
            value, spatial_shapes, level_start_index, sampling_locations, attention_weights,
            grad_output.contiguous(), ctx.im2col_step)
        return grad_value, None, None, grad_sampling_loc, grad_attn_weight, None
