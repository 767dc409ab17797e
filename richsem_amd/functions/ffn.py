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


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def pack_w2_bf16(w2):
    """linear2.weight (d_model, d_ffn) bf16 -> the hidden-column order the kernel reads (repack whenever the weight changes)."""
    if not w2.is_cuda:
        raise RuntimeError("Not implemented on the CPU")
    assert w2.dtype == torch.bfloat16 and w2.is_contiguous()
    out = torch.empty_like(w2)
    with torch.cuda.device(w2.device):
        _lib.check(_lib.load().msda_ffn_pack_w2_bf16(w2.data_ptr(), w2.shape[0], w2.shape[1], out.data_ptr(), _stream(w2)))
    return out


def ffn_forward_bf16(x, w1, b1, w2_packed, b2, ln_weight, ln_bias, eps=1e-5):
    """x (..., 256) bf16; w1 (d_ffn, 256) bf16; w2_packed from ``pack_w2_bf16``; biases / LayerNorm parameters float32."""
    if not x.is_cuda:
        raise RuntimeError("Not implemented on the CPU")
    assert x.dtype == torch.bfloat16 and w1.dtype == torch.bfloat16 and w2_packed.dtype == torch.bfloat16
    for t in (b1, b2, ln_weight, ln_bias):
        assert t.dtype == torch.float32 and t.is_contiguous()
    x2 = x.contiguous().view(-1, x.shape[-1])
    out = torch.empty_like(x2)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().msda_ffn_forward_bf16(
            x2.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2_packed.data_ptr(), b2.data_ptr(), ln_weight.data_ptr(),
            ln_bias.data_ptr(), float(eps), x2.shape[0], x2.shape[1], w1.shape[0], out.data_ptr(), _stream(x)))
    return out.view(x.shape)


class FusedFFNFunction(Function):
    """apply(x, w1, b1, w2, b2, ln_weight, ln_bias, eps): x, w1, w2 bf16; the rest float32."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, ln_weight, ln_bias, eps):
        out = ffn_forward_bf16(x, w1.contiguous(), b1, pack_w2_bf16(w2.contiguous()), b2, ln_weight, ln_bias, eps)
        ctx.save_for_backward(x, w1, b1, w2, b2, ln_weight, out)
        ctx.eps = eps
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        x, w1, b1, w2, b2, ln_weight, out = ctx.saved_tensors
        x2, g2 = x.reshape(-1, x.shape[-1]), grad_out.reshape(-1, x.shape[-1])
        # recompute the hidden activation (bf16 GEMM) and the LayerNorm input
        h = torch.relu(torch.addmm(b1.to(torch.bfloat16), x2, w1.t()))
        y = (x2.float() + torch.addmm(b2.to(torch.bfloat16), h, w2.t()).float())
        mean = y.mean(-1, keepdim=True)
        rstd = torch.rsqrt(y.var(-1, unbiased=False, keepdim=True) + ctx.eps)
        yhat = (y - mean) * rstd
        g = g2.float()
        grad_ln_w = (g * yhat).sum(0)
        grad_ln_b = g.sum(0)
        gy = g * ln_weight
        gy = (gy - gy.mean(-1, keepdim=True) - yhat * (gy * yhat).mean(-1, keepdim=True)) * rstd     # LayerNorm backward
        gyb = gy.to(torch.bfloat16)
        grad_b2 = gy.sum(0)
        grad_w2 = gyb.t() @ h
        gh = (gyb @ w2) * (h > 0)
        grad_b1 = gh.float().sum(0)
        grad_w1 = gh.t() @ x2
        grad_x = (gy + (gh @ w1).float()).to(torch.bfloat16).view(x.shape)
        return grad_x, grad_w1, grad_b1, grad_w2, grad_b2, grad_ln_w, grad_ln_b, None
