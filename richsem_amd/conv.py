"""Convolution forward on the matrix cores: csrc/conv_mfma.hip behind a small host API.

Activations are NHWC bfloat16 tensors of shape (N, H, W, C) (contiguous); a convolution comes with the per-channel affine that follows
it in the backbones (frozen / eval-mode BatchNorm folded into scale and shift -- reference models/richsem/backbone.py:20-56,
clip/model.py:16-27), an optional residual and an optional ReLU, all applied in the kernel's epilogue.  Forward only.
"""
import torch

from . import _lib


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def fold_bn(weight, bias, running_mean, running_var, eps=1e-5):
    """BatchNorm in eval mode / FrozenBatchNorm2d as y = x * scale + shift (backbone.py:45-56)"""
    scale = weight.float() * (running_var.float() + eps).rsqrt()
    return scale.contiguous(), (bias.float() - running_mean.float() * scale).contiguous()


def to_nhwc_bf16(x_nchw):
    return x_nchw.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)


def to_nchw(x_nhwc, dtype=torch.float32):
    return x_nhwc.permute(0, 3, 1, 2).to(dtype).contiguous()


class ConvAffine:
    """One convolution + affine (+ residual) (+ ReLU).  ``weight`` (C_out, C_in, KH, KW) as nn.Conv2d stores it; ``scale`` / ``shift``
    (C_out) fp32 (``None``: identity / zero).  C_out must be a multiple of 32.  Inputs with C_in not a multiple of 32 (the 3-channel
    stems, CLIP's 16-wide stem at small widths) go through explicit patches + a 1 x 1 product."""

    def __init__(self, weight, scale=None, shift=None, stride=1, padding=0, relu=False):
        if not weight.is_cuda:
            raise RuntimeError("Not implemented on the CPU")
        w = weight.detach().float().contiguous()
        self.Cout, self.Cin, self.KH, self.KW = w.shape
        assert self.Cout % 32 == 0, "C_out must be a multiple of 32"
        self.stride, self.pad, self.relu = int(stride), int(padding), bool(relu)
        dev = w.device
        self.scale = (torch.ones(self.Cout, device=dev) if scale is None else scale.detach().float().to(dev)).contiguous()
        self.shift = (torch.zeros(self.Cout, device=dev) if shift is None else shift.detach().float().to(dev)).contiguous()
        self.patches = self.Cin % 32 != 0
        if self.patches:      # k = (kh KW + kw) C_in + ci, padded to a multiple of 32: a 1 x 1 convolution over the patches
            k = self.KH * self.KW * self.Cin
            self.Kpad = (k + 31) // 32 * 32
            w2 = torch.zeros(self.Cout, self.Kpad, device=dev)
            w2[:, :k] = w.permute(0, 2, 3, 1).reshape(self.Cout, k)
            w = w2.view(self.Cout, self.Kpad, 1, 1).contiguous()
        self.packed = torch.empty(w.numel(), dtype=torch.int16, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().msda_conv_pack_weight(w.data_ptr(), w.shape[0], w.shape[1], w.shape[2], w.shape[3],
                                                         self.packed.data_ptr(), _stream(dev)))

    def out_hw(self, H, W):
        return (H + 2 * self.pad - self.KH) // self.stride + 1, (W + 2 * self.pad - self.KW) // self.stride + 1

    @torch.no_grad()
    def __call__(self, x, residual=None):
        """x (N, H, W, C_in) bf16 NHWC -> (N, Ho, Wo, C_out) bf16 NHWC"""
        if not x.is_cuda:
            raise RuntimeError("Not implemented on the CPU")
        assert x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[3] == self.Cin, (x.dtype, tuple(x.shape), self.Cin)
        x = x.contiguous()
        N, H, W, _ = x.shape
        Ho, Wo = self.out_hw(H, W)
        L = _lib.load()
        out = torch.empty((N, Ho, Wo, self.Cout), dtype=torch.bfloat16, device=x.device)
        if residual is not None:
            assert residual.shape == out.shape and residual.dtype == torch.bfloat16
            residual = residual.contiguous()
        with torch.cuda.device(x.device):
            st = _stream(x.device)
            if self.patches:
                pt = torch.empty((N, Ho, Wo, self.Kpad), dtype=torch.bfloat16, device=x.device)
                _lib.check(L.msda_conv_patches_bf16(x.data_ptr(), N, H, W, self.Cin, self.KH, self.KW, self.stride, self.pad, self.Kpad,
                                                    pt.data_ptr(), st))
                _lib.check(L.msda_conv_forward_bf16(pt.data_ptr(), self.packed.data_ptr(), self.scale.data_ptr(), self.shift.data_ptr(),
                                                    residual.data_ptr() if residual is not None else None, N, Ho, Wo, self.Kpad,
                                                    self.Cout, 1, 1, 1, 0, int(self.relu), out.data_ptr(), st))
            else:
                _lib.check(L.msda_conv_forward_bf16(x.data_ptr(), self.packed.data_ptr(), self.scale.data_ptr(), self.shift.data_ptr(),
                                                    residual.data_ptr() if residual is not None else None, N, H, W, self.Cin, self.Cout,
                                                    self.KH, self.KW, self.stride, self.pad, int(self.relu), out.data_ptr(), st))
        return out


def avg_pool_nhwc(x, k):
    """nn.AvgPool2d(k) on an NHWC tensor (clip/model.py:24, :36, :115): plain PyTorch on the channels-last view"""
    if k == 1:
        return x
    y = torch.nn.functional.avg_pool2d(x.permute(0, 3, 1, 2), k)
    return y.permute(0, 2, 3, 1).contiguous()
