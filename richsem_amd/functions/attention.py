"""Masked multi-head self-attention of the decoder's queries on the library's MFMA kernels (csrc/attn_mfma.hip; SURVEY.md section 8
row a9): what ``nn.MultiheadAttention`` computes between its input and output projections in
``DeformableTransformerDecoderLayer.forward_sa`` (reference models/richsem/deformable_transformer.py:974-978) --
``softmax(q k^T / sqrt(head_dim) + mask) v`` per (image, head) -- forward and backward, bf16 storage, fp32 softmax, head dimension 32,
sequence-first tensors.  The (nq, nq) score matrix never exists; the boolean mask (True = not allowed, the reference's convention)
travels as bits.  There is no CPU path.
"""
import ctypes

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib

_MASK_CACHE = {}


def mask_bits(mask):
    """(nq, nq) bool, True = masked -> (bits (nq, nkb) int32: bit j of word (q, kb) = mask[q, 32 kb + j]; the same of the transposed
    mask), kept for as long as the mask tensor is unchanged (the six layers of the decoder pass the same tensor).
    The cache entry HOLDS the mask tensor: while it does, the caching allocator cannot hand the mask's address to the next step's mask
    (``dn.prepare_dn_layout`` fills a fresh ``torch.empty`` buffer with a raw kernel, so ``_version`` is 0 for every mask it makes:
    with the address recycled, a different layout of equal shape would hit the stale bits).  A raw kernel that rewrites the SAME live
    buffer without going through torch is the one change this key cannot see."""
    key = (mask.data_ptr(), mask._version, tuple(mask.shape), mask.device)
    hit = _MASK_CACHE.get("last")
    if hit is not None and hit[0] == key and hit[2].untyped_storage().data_ptr() == mask.untyped_storage().data_ptr():
        return hit[1]
    assert mask.dtype == torch.bool and mask.dim() == 2 and mask.shape[0] == mask.shape[1]
    nq = mask.shape[0]
    nkb = (nq + 31) // 32
    w = (1 << torch.arange(32, device=mask.device, dtype=torch.int64))

    def to_i32(m):
        v = (torch.zeros((nq, nkb * 32), dtype=torch.int64, device=mask.device))
        v[:, :nq] = m
        s = (v.view(nq, nkb, 32) * w).sum(-1)
        s = torch.where(s >= 2 ** 31, s - 2 ** 32, s)
        return s.to(torch.int32).contiguous()

    val = (to_i32(mask), to_i32(mask.t()))
    _MASK_CACHE["last"] = (key, val, mask)      # (the tensor itself: keeps its storage, and so its address, taken)
    return val


def _ws(nq, bs, heads, device):
    n = _lib.load().msda_attn_workspace_bytes(nq, bs, heads)
    return torch.empty(n, dtype=torch.uint8, device=device)


class MaskedSelfAttentionFunction(Function):
    """apply(qk, v, mask, n_heads, batch_first): qk (nq, bs, 2 C) bf16 [or (bs, nq, 2 C) with ``batch_first``] -- queries in
    [..., :C], keys in [..., C:], the output of ONE stacked projection -- v the same with C channels, mask (nq, nq) bool (True =
    masked) or None -> the attention's output in v's layout"""

    @staticmethod
    def forward(ctx, qk, v, mask, n_heads, batch_first=False):
        if not qk.is_cuda:
            raise RuntimeError("Not implemented on the CPU")
        (bs, nq, c2) = qk.shape if batch_first else (qk.shape[1], qk.shape[0], qk.shape[2])
        C = c2 // 2
        assert C == n_heads * 32 and v.shape == qk.shape[:2] + (C,), "attention kernels: head dimension 32"
        qk, v = qk.contiguous(), v.contiguous()
        bits = mask_bits(mask) if mask is not None else (None, None)
        nqp = (nq + 31) // 32 * 32
        out = torch.empty_like(v)
        lse = torch.empty((bs * n_heads, nqp), dtype=torch.float32, device=qk.device)
        ws = _ws(nq, bs, n_heads, qk.device)
        esz = 2
        with _lib.on_device(qk.device):
            _lib.check(_lib.load().msda_attn_forward_bf16(
                qk.data_ptr(), c2, qk.data_ptr() + C * esz, c2, v.data_ptr(), C, bits[0].data_ptr() if bits[0] is not None else None,
                nq, bs, int(batch_first), n_heads, out.data_ptr(), lse.data_ptr(), ws.data_ptr(), _lib.raw_stream(qk.device)))
        ctx.save_for_backward(qk, v, out, lse, *(b for b in bits if b is not None))
        ctx.meta = (n_heads, mask is not None, batch_first, nq, bs)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        saved = ctx.saved_tensors
        qk, v, out, lse = saved[:4]
        n_heads, has_mask, batch_first, nq, bs = ctx.meta
        bits = saved[4:6] if has_mask else (None, None)
        c2 = qk.shape[2]
        C = c2 // 2
        dout = dout.contiguous()
        dqk, dv = torch.empty_like(qk), torch.empty_like(v)
        ws = _ws(nq, bs, n_heads, qk.device)
        with _lib.on_device(qk.device):
            _lib.check(_lib.load().msda_attn_backward_bf16(
                qk.data_ptr(), c2, qk.data_ptr() + C * 2, c2, v.data_ptr(), C, out.data_ptr(), dout.data_ptr(), lse.data_ptr(),
                bits[0].data_ptr() if has_mask else None, bits[1].data_ptr() if has_mask else None, nq, bs, int(batch_first), n_heads,
                dqk.data_ptr(), c2, dqk.data_ptr() + C * 2, c2, dv.data_ptr(), C, ws.data_ptr(),
                _lib.raw_stream(qk.device)))
        return dqk, dv, None, None, None


def masked_self_attention(qk, v, mask, n_heads, batch_first=False):
    return MaskedSelfAttentionFunction.apply(qk, v, mask, n_heads, batch_first)
