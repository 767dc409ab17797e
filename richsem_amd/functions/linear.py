"""Weight gradient of a linear layer on the matrix cores: dW = dY^T X with the contraction over the tokens -- the product the library's
transposed GEMM is slowest at (36 TFLOP/s at the MSDeformAttn projections' shape).  It is the convolution weight-gradient kernel
(csrc/conv_wgrad.hip) on a 1 x 1 convolution over a 1 x T "image": natural [token][channel] tiles staged through LDS and read
transposed.  Used by the bf16 module path for the four projections of MSDeformAttn (ops/modules/ms_deform_attn.py:52-56) and by the
feed-forward block's backward."""
import ctypes

import torch

from .. import _lib


def linear_wgrad_supported(out_features, in_features):
    return out_features % 128 == 0 and in_features % 128 == 0


def linear_wgrad_bf16(dy, x, with_bias=False):
    """dy (T, out_features), x (T, in_features), both bf16 and contiguous -> dW (out_features, in_features) float32; ``with_bias``: also
    the bias gradient sum_t dy[t] (out_features) float32, formed by the same kernel on the way"""
    assert dy.is_cuda and dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and dy.dim() == 2 and x.dim() == 2
    assert dy.shape[0] == x.shape[0]
    dy, x = dy.contiguous(), x.contiguous()
    T, cout = dy.shape
    cin = x.shape[1]
    L = _lib.load()
    dw = torch.empty((cout, cin), dtype=torch.float32, device=x.device)
    db = torch.empty(cout, dtype=torch.float32, device=x.device) if with_bias else None
    if T == 0:
        return (dw.zero_(), db.zero_()) if with_bias else dw.zero_()
    nb = ctypes.c_int64(0)
    _lib.check(L.msda_conv_wgrad_workspace_bytes(1, 1, T, cin, cout, 1, 1, 1, 0, ctypes.byref(nb)))
    ws = torch.empty(nb.value // 4, dtype=torch.float32, device=x.device) if nb.value else None
    with torch.cuda.device(x.device):
        _lib.check(L.msda_conv_wgrad_bf16(dy.data_ptr(), x.data_ptr(), 1, 1, T, cin, cout, 1, 1, 1, 0, dw.data_ptr(),
                                          db.data_ptr() if db is not None else None, None, 0, ws.data_ptr() if ws is not None else None,
                                          torch.cuda.current_stream(x.device).cuda_stream))
    return (dw, db) if with_bias else dw


class LinearBf16Function(torch.autograd.Function):
    """``F.linear`` on bf16 activations with fp32 parameters (cast per call): the forward and the input gradient are the library's bf16
    GEMMs, the weight gradient is :func:`linear_wgrad_bf16` (for layer sizes it supports and enough tokens to pay: the library's
    transposed GEMM otherwise), the bias gradient a column sum.  Gradients come back in the parameters' dtype."""

    MIN_TOKENS = 4096

    @staticmethod
    def forward(ctx, x, weight, bias):
        w16 = weight.to(torch.bfloat16)
        ctx.save_for_backward(x, w16)
        ctx.meta = (weight.dtype, bias.dtype if bias is not None else None)
        # written into a tensor of the final shape (F.linear on a 3-d input returns a view, which a custom Function must not hand out
        # when the caller may modify it in place -- the module's padding mask does)
        out = torch.empty(x.shape[:-1] + (w16.shape[0],), dtype=torch.bfloat16, device=x.device)
        x2 = x.reshape(-1, x.shape[-1])
        if bias is not None:
            torch.addmm(bias.to(torch.bfloat16), x2, w16.t(), out=out.view(-1, w16.shape[0]))
        else:
            torch.mm(x2, w16.t(), out=out.view(-1, w16.shape[0]))
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, w16 = ctx.saved_tensors
        wdt, bdt = ctx.meta
        dy2, x2 = dy.reshape(-1, dy.shape[-1]).contiguous(), x.reshape(-1, x.shape[-1])
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = (dy2 @ w16).view(x.shape)
        want_b = bdt is not None and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            if linear_wgrad_supported(dy2.shape[1], x2.shape[1]) and dy2.shape[0] >= LinearBf16Function.MIN_TOKENS:
                if want_b:
                    dw, db = linear_wgrad_bf16(dy2, x2.contiguous(), with_bias=True)
                    db = db.to(bdt)
                else:
                    dw = linear_wgrad_bf16(dy2, x2.contiguous())
                dw = dw.to(wdt)
            else:
                dw = (dy2.t() @ x2).to(wdt)
        if want_b and db is None:
            db = dy2.sum(0, dtype=torch.float32).to(bdt)
        return dx, dw, db


def linear_bf16(x, weight, bias=None):
    return LinearBf16Function.apply(x, weight, bias)


def lin256_pack(weight):
    """weight (out_features, 256) -> bf16 in the fragment order of ``lin256`` (csrc/lin256_mfma.hip); out_features % 64 == 0"""
    w = weight.detach().to(torch.bfloat16).contiguous()
    assert w.is_cuda and w.dim() == 2 and w.shape[1] == 256 and w.shape[0] % 64 == 0
    packed = torch.empty_like(w)
    with torch.cuda.device(w.device):
        _lib.check(_lib.load().msda_lin256_pack_bf16(w.data_ptr(), w.shape[0], 256, packed.data_ptr(),
                                                     torch.cuda.current_stream(w.device).cuda_stream))
    return packed


def lin256(x, packed_w, bias=None, relu=False, relu_mask=None):
    """x (T, 256) bf16 -> (T, out_features) bf16: ``x W^T + bias`` (``relu``: with ReLU), or ``(x W^T) * (relu_mask > 0)`` when
    ``relu_mask`` (T, out_features) bf16 is given (the gradient at a ReLU's input from the gradient at its output)"""
    assert x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 2 and x.shape[1] == 256
    x = x.contiguous()
    N = packed_w.shape[0]
    out = torch.empty((x.shape[0], N), dtype=torch.bfloat16, device=x.device)
    if relu_mask is not None:
        assert relu_mask.shape == out.shape and relu_mask.dtype == torch.bfloat16 and relu_mask.is_contiguous() and bias is None
    epi = 2 if relu_mask is not None else (1 if relu else 0)
    b = bias.detach().float().contiguous() if bias is not None else None
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().msda_lin256_forward_bf16(x.data_ptr(), packed_w.data_ptr(), b.data_ptr() if b is not None else None,
                                                        relu_mask.data_ptr() if relu_mask is not None else None, epi, x.shape[0], 256, N,
                                                        out.data_ptr(), torch.cuda.current_stream(x.device).cuda_stream))
    return out


def lin256_f32_pack(weight):
    """weight (out_features, 256) float32 -> bf16 hi + lo parts in the fragment order of ``lin256_f32``; out_features % 32 == 0"""
    w = weight.detach().float().contiguous()
    assert w.is_cuda and w.dim() == 2 and w.shape[1] == 256 and w.shape[0] % 32 == 0
    packed = torch.empty(2 * w.numel(), dtype=torch.int16, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(_lib.load().msda_lin256_pack_f32(w.data_ptr(), w.shape[0], 256, packed.data_ptr(),
                                                    torch.cuda.current_stream(w.device).cuda_stream))
    return packed


def lin256_f32(x, packed_w, out_features, bias=None):
    """x (T, 256) float32 -> x W^T + bias (T, out_features) float32 at fp32-level accuracy on bf16 MFMAs (three per tile)"""
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] == 256
    x = x.contiguous()
    out = torch.empty((x.shape[0], out_features), dtype=torch.float32, device=x.device)
    b = bias.detach().float().contiguous() if bias is not None else None
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().msda_lin256_forward_f32(x.data_ptr(), packed_w.data_ptr(), b.data_ptr() if b is not None else None,
                                                       x.shape[0], 256, out_features, out.data_ptr(),
                                                       torch.cuda.current_stream(x.device).cuda_stream))
    return out


class LinearBf16CachedFunction(torch.autograd.Function):
    """:class:`LinearBf16Function` for a caller that keeps the bf16 casts of its parameters (and the concatenation of several layers'
    parameters into one projection) across calls: ``apply(x, w16, b16, split, *params)`` computes ``x w16^T + b16`` and routes the
    gradients to ``params`` = (weight, bias) or, with ``split`` = rows of the first layer, (weight_a, weight_b, bias_a, bias_b) -- the
    two layers whose parameters ``w16`` / ``b16`` concatenate.  No cast or concatenation kernels per call."""

    @staticmethod
    def forward(ctx, x, w16, b16, split, *params):
        ctx.save_for_backward(x, w16)
        ctx.meta = (split, tuple(p.dtype for p in params))
        out = torch.empty(x.shape[:-1] + (w16.shape[0],), dtype=torch.bfloat16, device=x.device)
        torch.addmm(b16, x.reshape(-1, x.shape[-1]), w16.t(), out=out.view(-1, w16.shape[0]))
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, w16 = ctx.saved_tensors
        split, dts = ctx.meta
        dy2, x2 = dy.reshape(-1, dy.shape[-1]).contiguous(), x.reshape(-1, x.shape[-1])
        dx = (dy2 @ w16).view(x.shape) if ctx.needs_input_grad[0] else None
        need = ctx.needs_input_grad[4:]
        if not any(need):          # frozen projections, or only the input gradient is wanted: no token contraction at all
            return (dx, None, None, None) + (None,) * len(dts)
        nw = 1 if split is None else 2
        want_w, want_b = any(need[:nw]), any(need[nw:])
        dw = db = None
        if want_w and linear_wgrad_supported(dy2.shape[1], x2.shape[1]) and dy2.shape[0] >= LinearBf16Function.MIN_TOKENS:
            if want_b:
                dw, db = linear_wgrad_bf16(dy2, x2.contiguous(), with_bias=True)
            else:
                dw = linear_wgrad_bf16(dy2, x2.contiguous())
        elif want_w:
            dw = (dy2.t() @ x2).float()
        if want_b and db is None:
            db = dy2.sum(0, dtype=torch.float32)
        if split is None:
            grads = (dw, db)
        else:
            grads = (dw[:split] if dw is not None else None, dw[split:] if dw is not None else None,
                     db[:split] if db is not None else None, db[split:] if db is not None else None)
        return (dx, None, None, None) + tuple(g.to(dt) if g is not None and n else None for g, dt, n in zip(grads, dts, need))
