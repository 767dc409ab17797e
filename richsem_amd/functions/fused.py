"""Fused module path: what the reference module does with half a dozen PyTorch ops around the
operator (reference models/richsem/ops/modules/ms_deform_attn.py:94-109) as ONE kernel each way, plus an in-place padding
mask -- kernels in richsem_amd/csrc/msda_prep.h, C ABI ``msda_prep_forward_* / msda_prep_backward_* / msda_mask_rows_*``.

``MSDeformAttnFusedFunction.apply(value, spatial_shapes, level_start_index, qproj, reference_points, n_heads, n_levels,
n_points, im2col_step)`` takes the RAW output of the offset and attention-weight projections -- one tensor
(N, Lq, M*L*P*2 + M*L*P), i.e. one 256 -> 384 GEMM for RichSem -- and returns the operator's output; its backward returns the
gradient of that one tensor (one GEMM backward), of value and of the reference points.
"""
import ctypes

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import MultiScaleDeformableAttention as MSDA
from .. import _lib

_SFX = {torch.float32: "f32", torch.float64: "f64", torch.bfloat16: "bf16"}


def _stream(t):
    return _lib.raw_stream(t.device)


class MaskRows(Function):
    """value.masked_fill(mask[..., None], 0) IN PLACE: only the rows of padded pixels are touched (reference :95-96)."""

    @staticmethod
    def forward(ctx, value, mask):
        assert value.is_contiguous() and mask.dtype == torch.bool
        mask = mask.contiguous()
        sfx = {torch.float32: "f32", torch.float64: "f64", torch.bfloat16: "bf16"}[value.dtype]
        rows = mask.numel()
        with _lib.on_device(value.device):
            _lib.check(getattr(_lib.load(), "msda_mask_rows_" + sfx)(
                value.data_ptr(), mask.view(torch.uint8).data_ptr(), rows, value.numel() // rows, _stream(value)))
        ctx.mark_dirty(value)
        ctx.save_for_backward(mask)
        return value

    @staticmethod
    @once_differentiable
    def backward(ctx, grad):
        (mask,) = ctx.saved_tensors
        grad = grad.contiguous().clone()
        sfx = {torch.float32: "f32", torch.float64: "f64", torch.bfloat16: "bf16"}[grad.dtype]
        rows = mask.numel()
        with _lib.on_device(grad.device):
            _lib.check(getattr(_lib.load(), "msda_mask_rows_" + sfx)(
                grad.data_ptr(), mask.view(torch.uint8).data_ptr(), rows, grad.numel() // rows, _stream(grad)))
        return grad, None


class MSDeformAttnFusedFunction(Function):
    @staticmethod
    def forward(ctx, value, spatial_shapes, level_start_index, qproj, reference_points, n_heads, n_levels, n_points,
                im2col_step):
        if not value.is_cuda:
            raise RuntimeError("Not implemented on the CPU")
        # bfloat16: value and the raw projection are bf16 (bf16 GEMMs around the operator); reference points, locations and
        # attention weights are float32, arithmetic fp32 (msda_prep_*_bf16, msda_forward_bf16 / msda_backward_bf16)
        work = torch.float32 if qproj.dtype == torch.bfloat16 else qproj.dtype
        if qproj.dtype not in _SFX or reference_points.dtype != work:
            raise RuntimeError(f"fused MSDeformAttn path: float32 / float64 / bfloat16 projections with reference points in "
                               f"{work}, got {qproj.dtype} / {reference_points.dtype}")
        N, Lq = qproj.shape[0], qproj.shape[1]
        M, L, P = n_heads, n_levels, n_points
        n_off, n_log = M * L * P * 2, M * L * P
        if qproj.shape[-1] != n_off + n_log or not qproj.is_contiguous():
            raise RuntimeError("qproj must be a contiguous (N, Lq, M*L*P*3) tensor: offsets, then attention logits")
        ref = reference_points.contiguous()
        if ref.shape[:3] != (N, Lq, L) or ref.shape[-1] not in (2, 4):
            raise ValueError(f"Last dim of reference_points must be 2 or 4, but get {ref.shape[-1]} instead.")
        sh, ls = MSDA._host_mirror(spatial_shapes, level_start_index)
        lib = _lib.load()
        loc = torch.empty((N, Lq, M, L, P, 2), dtype=work, device=qproj.device)
        aw = torch.empty((N, Lq, M, L, P), dtype=work, device=qproj.device)
        stride, esz = qproj.shape[-1], qproj.element_size()
        # ONE entry point for softmax + location arithmetic + gather (msda_forward_prep_*): a single kernel for decoder-shaped calls, the
        # location / softmax kernel followed by the forward for the others; loc / aw come back as by-products for the backward
        value = value.contiguous()
        if (value.dtype != qproj.dtype or value.dim() != 4 or value.shape[0] != N or value.shape[2] != M
                or spatial_shapes.dtype != torch.int64 or level_start_index.dtype != torch.int64):
            raise RuntimeError("fused MSDeformAttn path: value must be a (N, S, M, D) tensor of the projection's dtype, shapes int64")
        S, D = value.shape[1], value.shape[3]
        out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
        with _lib.on_device(qproj.device):
            _lib.check(getattr(lib, "msda_forward_prep_" + _SFX[qproj.dtype])(
                value.data_ptr(), spatial_shapes.contiguous().data_ptr(), level_start_index.contiguous().data_ptr(),
                qproj.data_ptr(), stride, qproj.data_ptr() + n_off * esz, stride, ref.data_ptr(), ref.shape[-1],
                N, S, M, D, L, Lq, P, im2col_step, out.data_ptr(), loc.data_ptr(), aw.data_ptr(), sh.ctypes.data, ls.ctypes.data,
                _stream(qproj)))
        ctx.save_for_backward(value, spatial_shapes, level_start_index, loc, aw, ref, qproj)
        ctx.dims = (M, L, P, im2col_step)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, spatial_shapes, level_start_index, loc, aw, ref, qproj = ctx.saved_tensors
        M, L, P, step = ctx.dims
        grad_value, grad_loc, grad_aw = MSDA.ms_deform_attn_backward(
            value, spatial_shapes, level_start_index, loc, aw, grad_output.contiguous(), step)
        N, Lq = qproj.shape[0], qproj.shape[1]
        n_off = M * L * P * 2
        sh, _ = MSDA._host_mirror(spatial_shapes, level_start_index)
        grad_qproj = torch.empty_like(qproj)
        grad_ref = torch.empty_like(ref) if ctx.needs_input_grad[4] else None
        stride, esz = qproj.shape[-1], qproj.element_size()
        with _lib.on_device(qproj.device):
            _lib.check(getattr(_lib.load(), "msda_prep_backward_" + _SFX[qproj.dtype])(
                grad_loc.data_ptr(), grad_aw.data_ptr(), aw.data_ptr(), qproj.data_ptr(), stride, ref.data_ptr(),
                ref.shape[-1], sh.ctypes.data, N, Lq, M, L, P, grad_qproj.data_ptr(), stride,
                grad_qproj.data_ptr() + n_off * esz, stride,
                ctypes.c_void_p(grad_ref.data_ptr()) if grad_ref is not None else None, _stream(qproj)))
        return grad_value, None, None, grad_qproj, grad_ref, None, None, None, None
