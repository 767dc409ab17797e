"""The transformer layers' feed-forward block on the MI355X matrix cores:

    out = LayerNorm(x + linear2(relu(linear1(x))))        reference models/richsem/deformable_transformer.py:862-866, :940-944

``ffn_forward_bf16`` is the one-kernel forward (C ABI ``msda_ffn_forward_bf16``, kernel richsem_amd/csrc/ffn_mfma.hip): bf16
storage, fp32 accumulation, the hidden activation never leaves the chip.  ``FusedFFNFunction`` makes it differentiable: the
backward recomputes the hidden activation and runs as plain bf16 GEMMs (library GEMMs) + element-wise work.
There is no CPU path: CPU tensors raise.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib


# The one-kernel forward walks all d_ffn / 32 weight tiles in every workgroup of 192 tokens: below this many tokens too few
# CUs are busy and the block as library GEMMs is faster (MI355X: 2200 tokens 97 vs 78 us, 8000 tokens 100 vs 78 us, 44646 tokens
# 117 vs 290 us).  The modules fall back to the op-by-op sequence below it.
FUSED_FFN_MIN_TOKENS = 16384


def _wgrad(dy, x, with_bias=False, fp32=False):
    """dW = dy^T x (contraction over the tokens): the library's own MFMA kernel where it applies (functions/linear.py: 131 vs
    208 us at the encoder shape), else the library's transposed GEMM.  bf16 result unless ``fp32`` (the kernel's own sums, for a caller
    that hands them to float32 master parameters: no cast to bf16 and back)"""
    from .linear import LinearBf16Function, linear_wgrad_bf16, linear_wgrad_supported
    if linear_wgrad_supported(dy.shape[1], x.shape[1]) and dy.shape[0] >= LinearBf16Function.MIN_TOKENS:
        if with_bias:
            dw, db = linear_wgrad_bf16(dy, x.contiguous(), with_bias=True)
            return (dw if fp32 else dw.to(torch.bfloat16)), db
        dw = linear_wgrad_bf16(dy, x.contiguous())
        return dw if fp32 else dw.to(torch.bfloat16)
    dw = dy.t() @ x
    return (dw, dy.sum(0, dtype=torch.float32)) if with_bias else dw


def _stream(t):
    return _lib.raw_stream(t.device)


def pack_w2_bf16(w2):
    """linear2.weight (d_model, d_ffn) bf16 -> the hidden-column order the kernel reads (repack whenever the weight changes)."""
    if not w2.is_cuda:
        raise RuntimeError("Not implemented on the CPU")
    assert w2.dtype == torch.bfloat16 and w2.is_contiguous()
    out = torch.empty_like(w2)
    with _lib.on_device(w2.device):
        _lib.check(_lib.load().msda_ffn_pack_w2_bf16(w2.data_ptr(), w2.shape[0], w2.shape[1], out.data_ptr(), _stream(w2)))
    return out


def ffn_forward_bf16(x, w1, b1, w2_packed, b2, ln_weight, ln_bias, eps=1e-5, return_rstd=False):
    """x (..., 256) bf16; w1 (d_ffn, 256) bf16; w2_packed from ``pack_w2_bf16``; biases / LayerNorm parameters float32.
    ``return_rstd``: also the LayerNorm's 1 / sqrt(var + eps) per token (float32) and its normalised input yhat (bf16), which the
    backward needs."""
    if not x.is_cuda:
        raise RuntimeError("Not implemented on the CPU")
    assert x.dtype == torch.bfloat16 and w1.dtype == torch.bfloat16 and w2_packed.dtype == torch.bfloat16
    for t in (b1, b2, ln_weight, ln_bias):
        assert t.dtype == torch.float32 and t.is_contiguous()
    x2 = x.contiguous().view(-1, x.shape[-1])
    out = torch.empty_like(x2)
    rstd = torch.empty(x2.shape[0], dtype=torch.float32, device=x.device) if return_rstd else None
    yhat = torch.empty_like(x2) if return_rstd else None
    with _lib.on_device(x.device):
        _lib.check(_lib.load().msda_ffn_forward_train_bf16(
            x2.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2_packed.data_ptr(), b2.data_ptr(), ln_weight.data_ptr(),
            ln_bias.data_ptr(), float(eps), x2.shape[0], x2.shape[1], w1.shape[0], out.data_ptr(),
            rstd.data_ptr() if rstd is not None else None, yhat.data_ptr() if yhat is not None else None, _stream(x)))
    return (out.view(x.shape), rstd, yhat) if return_rstd else out.view(x.shape)


def ffn_ln_backward_bf16(grad_out, yhat, rstd, ln_weight):
    """First step of the backward (``msda_ffn_ln_backward_bf16``): -> (dz bf16 (tokens, 256), grad_ln_weight, grad_ln_bias, grad_b2)"""
    g2, o2 = grad_out.contiguous().view(-1, 256), yhat.contiguous().view(-1, 256)
    dz = torch.empty_like(o2)
    sums = torch.empty(3, 256, dtype=torch.float32, device=o2.device)
    with _lib.on_device(o2.device):
        _lib.check(_lib.load().msda_ffn_ln_backward_bf16(g2.data_ptr(), o2.data_ptr(), rstd.data_ptr(), ln_weight.data_ptr(),
                                                         o2.shape[0], 256, dz.data_ptr(), sums[0].data_ptr(),
                                                         sums[1].data_ptr(), sums[2].data_ptr(), _stream(o2)))
    return dz, sums[0], sums[1], sums[2]


class FusedFFNFunction(Function):
    """apply(x, w1, b1, w2, b2, ln_weight, ln_bias, eps): x, w1, w2 bf16; the rest float32."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, ln_weight, ln_bias, eps):
        out, rstd, yhat = ffn_forward_bf16(x, w1.contiguous(), b1, pack_w2_bf16(w2.contiguous()), b2, ln_weight, ln_bias, eps,
                                           return_rstd=True)
        ctx.save_for_backward(x, w1, b1, w2, ln_weight, yhat, rstd)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        x, w1, b1, w2, ln_weight, yhat, rstd = ctx.saved_tensors
        x2 = x.reshape(-1, x.shape[-1])
        # LayerNorm backward + the three token sums in one kernel (yhat and rstd stored by the forward)
        dz, grad_ln_w, grad_ln_b, grad_b2 = ffn_ln_backward_bf16(grad_out.to(torch.bfloat16), yhat, rstd, ln_weight)
        # the two token-parallel products with K = 256 on the library's own kernel (csrc/lin256_mfma.hip: 81 / 114 us against 173 / 204 us
        # for the library's GEMM + element-wise op): the hidden activation, recomputed, and the gradient at the ReLU's input
        from .linear import lin256, lin256_pack
        if w1.shape[0] % 64 == 0:
            h = lin256(x2, lin256_pack(w1), b1, relu=True)
            grad_w2 = _wgrad(dz, h)
            gh = lin256(dz, lin256_pack(w2.t()), relu_mask=h)
        else:
            h = torch.relu(torch.addmm(b1.to(torch.bfloat16), x2, w1.t()))
            grad_w2 = _wgrad(dz, h)
            gh = torch.ops.aten.threshold_backward(dz @ w2, h, 0)
        grad_w1, grad_b1 = _wgrad(gh, x2, with_bias=True)
        grad_x = torch.addmm(dz, gh, w1).view(x.shape)                          # residual + first product's input gradient
        return grad_x, grad_w1, grad_b1, grad_w2, grad_b2, grad_ln_w, grad_ln_b, None


def pack_ffn(linear1_weight, linear1_bias, linear2_weight):
    """The derived forms :class:`FusedFFNCachedFunction` runs the feed-forward block from (kept by the layer in a ``VersionCache``: rebuilt
    when a parameter changes, not per call and not again in the backward): linear1.weight as bf16 and in lin256's fragment order,
    linear2.weight in the fused kernel's hidden-column order and its transpose in lin256's order, linear1.bias as float32."""
    from .linear import lin256_pack
    w1_16 = linear1_weight.detach().to(torch.bfloat16).contiguous()
    w2_16 = linear2_weight.detach().to(torch.bfloat16).contiguous()
    return {"w1_16": w1_16, "w1_packed": lin256_pack(w1_16), "w2_ffn": pack_w2_bf16(w2_16), "w2t_packed": lin256_pack(w2_16.t().contiguous()),
            "b1_32": linear1_bias.detach().float().contiguous()}


class FusedFFNCachedFunction(Function):
    """:class:`FusedFFNFunction` for a caller that keeps the packed parameters (``pk`` from :func:`pack_ffn`) across calls and hands over the
    float32 master parameters themselves: ``apply(x, pk, eps, linear1.weight, linear1.bias, linear2.weight, linear2.bias, norm.weight,
    norm.bias)``.  No cast / pack / transpose kernels per call (the uncached form runs three in the forward and four in the backward), and
    the weight gradients reach the parameters in float32 as the kernel sums them -- not rounded to bf16 on the way through a cast node."""

    @staticmethod
    def forward(ctx, x, pk, eps, w1, b1, w2, b2, ln_weight, ln_bias):
        lw, lb = ln_weight.detach().float().contiguous(), ln_bias.detach().float().contiguous()
        out, rstd, yhat = ffn_forward_bf16(x, pk["w1_16"], pk["b1_32"], pk["w2_ffn"], b2.detach().float().contiguous(), lw, lb, eps,
                                           return_rstd=True)
        ctx.save_for_backward(x, lw, yhat, rstd)
        ctx.pk, ctx.dts = pk, tuple(p.dtype for p in (w1, b1, w2, b2, ln_weight, ln_bias))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        from .linear import lin256, linear_wgrad_bf16
        x, lw, yhat, rstd = ctx.saved_tensors
        pk, dts = ctx.pk, ctx.dts
        x2 = x.reshape(-1, x.shape[-1])
        dz, g_lnw, g_lnb, g_b2 = ffn_ln_backward_bf16(grad_out.to(torch.bfloat16), yhat, rstd, lw)
        h = lin256(x2, pk["w1_packed"], pk["b1_32"], relu=True)                 # the hidden activation, recomputed
        need = ctx.needs_input_grad
        g_w2 = linear_wgrad_bf16(dz, h).to(dts[2]) if need[5] else None
        gh = lin256(dz, pk["w2t_packed"], relu_mask=h)                           # gradient at the ReLU's input
        g_w1 = g_b1 = None
        if need[3] or need[4]:
            g_w1, g_b1 = linear_wgrad_bf16(gh, x2.contiguous(), with_bias=True)
            g_w1, g_b1 = g_w1.to(dts[0]), g_b1.to(dts[1])
        dx = torch.addmm(dz, gh, pk["w1_16"]).view(x.shape) if need[0] else None
        return dx, None, None, g_w1, g_b1, g_w2, g_b2.to(dts[3]), g_lnw.to(dts[4]), g_lnb.to(dts[5])


def add_layernorm_forward_bf16(a2, b2, ln_weight32, ln_bias32, eps, need_backward=True):
    """LayerNorm(a2 + b2) over 256 channels, (tokens, 256) bf16 -> out, and (for the backward) rstd (tokens) f32, yhat (tokens, 256) bf16"""
    out = torch.empty_like(a2)
    rstd = torch.empty(a2.shape[0], dtype=torch.float32, device=a2.device) if need_backward else None
    yhat = torch.empty_like(a2) if need_backward else None
    with _lib.on_device(a2.device):
        _lib.check(_lib.load().msda_add_layernorm_forward_bf16(
            a2.data_ptr(), b2.data_ptr(), ln_weight32.data_ptr(), ln_bias32.data_ptr(), float(eps), a2.shape[0], 256, out.data_ptr(),
            rstd.data_ptr() if need_backward else None, yhat.data_ptr() if need_backward else None, _stream(a2)))
    return out, rstd, yhat


class FFNSmallFunction(Function):
    """The feed-forward block ``LayerNorm(x + linear2(relu(linear1(x))))`` for FEW tokens (the decoder's ~2 k queries, where the
    one-kernel forward of :class:`FusedFFNFunction` leaves most CUs idle): first product with bias + ReLU on ``lin256``, second
    product (K = d_ffn) on the library's bf16 GEMM, residual + LayerNorm as one kernel; backward: LayerNorm gradient + token sums in
    one kernel, the gradient at the ReLU's input on ``lin256`` (mask epilogue), weight gradients as transposed GEMMs.
    ``apply(x, pk1, w2_16, w2t_packed, eps, w1, b1, w2, b2, ln_weight, ln_bias)``: ``pk1`` = pack_linear256 of linear1, ``w2_16`` =
    linear2.weight in bf16, ``w2t_packed`` = lin256_pack of its transpose (kept by the caller); the last six are the parameters the
    gradients go to."""

    @staticmethod
    def forward(ctx, x, pk1, w2_16, w2t_packed, eps, w1, b1, w2, b2, ln_weight, ln_bias):
        from .linear import lin256
        x2 = x.reshape(-1, 256).contiguous()
        h = lin256(x2, pk1["packed"], pk1["b32"], relu=True)
        y = torch.addmm(b2.detach().to(torch.bfloat16), h, w2_16.t())
        lw, lb = ln_weight.detach().float().contiguous(), ln_bias.detach().float().contiguous()
        out, rstd, yhat = add_layernorm_forward_bf16(x2, y, lw, lb, eps)
        ctx.save_for_backward(x2, h, yhat, rstd, lw, w2t_packed, pk1["w16"])
        ctx.meta = (x.shape, tuple(p.dtype for p in (w1, b1, w2, b2, ln_weight, ln_bias)))
        from .linear import WgradGroup
        ctx.group = WgradGroup.active_for((w1, b1, w2))      # (only a boundary's aliases are deferred: functions/linear.py)
        return out.view(x.shape)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        from .linear import lin256
        x2, h, yhat, rstd, lw, w2t_packed, w1_16 = ctx.saved_tensors
        shape, dts = ctx.meta
        dz, g_lnw, g_lnb, g_b2 = ffn_ln_backward_bf16(grad_out.to(torch.bfloat16), yhat, rstd, lw)
        need = ctx.needs_input_grad
        from .linear import deferrable
        grp2 = deferrable(ctx.group, dz, h, (dts[2],))      # (the layer's deferred weight gradients: functions/linear.py, WgradGroup)
        if need[7]:
            g_w2 = grp2.add(dz, h, False)[0] if grp2 is not None else _wgrad(dz, h, fp32=dts[2] == torch.float32).to(dts[2])
        else:
            g_w2 = None
        gh = lin256(dz, w2t_packed, relu_mask=h)
        g_w1 = g_b1 = None
        if need[5] or need[6]:
            grp1 = deferrable(ctx.group, gh, x2, (dts[0], dts[1])) if (need[5] and need[6]) else None
            if grp1 is not None:
                g_w1, g_b1 = grp1.add(gh, x2, True)
            else:
                g_w1, g_b1 = _wgrad(gh, x2, with_bias=True, fp32=dts[0] == torch.float32)
                g_w1, g_b1 = g_w1.to(dts[0]), g_b1.to(dts[1])
        dx = torch.addmm(dz, gh, w1_16).view(shape) if need[0] else None
        return dx, None, None, None, None, g_w1, g_b1, g_w2, g_b2.to(dts[3]), g_lnw.to(dts[4]), g_lnb.to(dts[5])


class AddLayerNormFunction(Function):
    """``LayerNorm(a + b)`` over 256 channels in bf16 (the layers' norm1 around the attention's residual, reference
    deformable_transformer.py:876-877 with the dropout inactive) as one kernel forward (``msda_add_layernorm_forward_bf16``) and one
    backward (``msda_ffn_ln_backward_bf16``): apply(a, b, ln_weight, ln_bias, eps); a, b bf16, the parameters float32."""

    @staticmethod
    def forward(ctx, a, b, ln_weight, ln_bias, eps):
        a2, b2 = a.contiguous().view(-1, 256), b.contiguous().view(-1, 256)
        out = torch.empty_like(a2)
        need = any(ctx.needs_input_grad)
        rstd = torch.empty(a2.shape[0], dtype=torch.float32, device=a.device) if need else None
        yhat = torch.empty_like(a2) if need else None
        w, bi = ln_weight.detach().float().contiguous(), ln_bias.detach().float().contiguous()
        with _lib.on_device(a.device):
            _lib.check(_lib.load().msda_add_layernorm_forward_bf16(
                a2.data_ptr(), b2.data_ptr(), w.data_ptr(), bi.data_ptr(), float(eps), a2.shape[0], 256, out.data_ptr(),
                rstd.data_ptr() if need else None, yhat.data_ptr() if need else None, _stream(a)))
        if need:
            ctx.save_for_backward(w, yhat, rstd)
        ctx.meta = (a.shape, ln_weight.dtype, ln_bias.dtype)
        return out.view(a.shape)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        w, yhat, rstd = ctx.saved_tensors
        shape, wdt, bdt = ctx.meta
        dz, gw, gb, _ = ffn_ln_backward_bf16(grad_out.to(torch.bfloat16), yhat, rstd, w)
        dz = dz.view(shape)
        return dz, dz, gw.to(wdt), gb.to(bdt), None
