"""Convolution forward on the matrix cores: csrc/conv_mfma.hip behind a small host API.

Activations are NHWC bfloat16 tensors of shape (N, H, W, C) (contiguous); a convolution comes with the per-channel affine that follows
it in the backbones (frozen / eval-mode BatchNorm folded into scale and shift -- reference models/richsem/backbone.py:20-56,
clip/model.py:16-27), an optional residual and an optional ReLU, all applied in the kernel's epilogue.  ``ConvAffine`` is the inference form;
``ConvAffineFunction`` / ``ConvBNAct`` add the input and weight gradients.
"""
import ctypes

import torch

from . import _lib


def _stream(dev):
    return _lib.raw_stream(dev)


def fold_bn(weight, bias, running_mean, running_var, eps=1e-5):
    """BatchNorm in eval mode / FrozenBatchNorm2d as y = x * scale + shift (backbone.py:45-56)"""
    scale = weight.float() * (running_var.float() + eps).rsqrt()
    return scale.contiguous(), (bias.float() - running_mean.float() * scale).contiguous()


def to_nhwc_bf16(x_nchw):
    return x_nchw.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)


def to_nchw(x_nhwc, dtype=torch.float32):
    return x_nhwc.permute(0, 3, 1, 2).to(dtype).contiguous()


_WS_BYTES = {}      # (entry point, problem dimensions) -> bytes: the library's answer depends on nothing else


def _ksplit_workspace(fn, dims, device):
    """zeroed fp32 workspace for a k-split convolution call, or None where the library does not split the problem"""
    key = (fn.__name__, dims)
    n = _WS_BYTES.get(key)
    if n is None:
        nb = ctypes.c_int64(0)
        _lib.check(fn(*dims, ctypes.byref(nb)))
        n = _WS_BYTES[key] = nb.value
    return torch.zeros(n // 4, dtype=torch.float32, device=device) if n else None


class ConvAffine:
    """One convolution + affine (+ residual) (+ ReLU).  ``weight`` (C_out, C_in, KH, KW) as nn.Conv2d stores it; ``scale`` / ``shift``
    (C_out) fp32 (``None``: identity / zero).  C_out must be a multiple of 16; C_in a multiple of 32, or KH KW C_in <= 512 (the
    3-channel stems)."""

    flop_counter = None      # set to [0.0] to add up 2 * MACs of the calls that follow (bench.py's MFMA rows)

    def __init__(self, weight, scale=None, shift=None, stride=1, padding=0, relu=False):
        if not weight.is_cuda:
            raise RuntimeError("Not implemented on the CPU")
        w = weight.detach().float().contiguous()
        self.Cout, self.Cin, self.KH, self.KW = w.shape
        assert self.Cout % 16 == 0, "C_out must be a multiple of 16"
        self.stride, self.pad, self.relu = int(stride), int(padding), bool(relu)
        dev = w.device
        self.scale = (torch.ones(self.Cout, device=dev) if scale is None else scale.detach().float().to(dev)).contiguous()
        self.shift = (torch.zeros(self.Cout, device=dev) if shift is None else shift.detach().float().to(dev)).contiguous()
        L = _lib.load()
        n = ctypes.c_int64(0)
        _lib.check(L.msda_conv_packed_elems(self.Cout, self.Cin, self.KH, self.KW, ctypes.byref(n)))
        self.packed = torch.empty(n.value, dtype=torch.int16, device=dev)
        with _lib.on_device(dev):
            _lib.check(L.msda_conv_pack_weight(w.data_ptr(), self.Cout, self.Cin, self.KH, self.KW, self.packed.data_ptr(), _stream(dev)))

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
        if ConvAffine.flop_counter is not None:
            ConvAffine.flop_counter[0] += 2.0 * N * Ho * Wo * self.Cout * self.Cin * self.KH * self.KW
        out = torch.empty((N, Ho, Wo, self.Cout), dtype=torch.bfloat16, device=x.device)
        if residual is not None:
            assert residual.shape == out.shape and residual.dtype == torch.bfloat16
            residual = residual.contiguous()
        with _lib.on_device(x.device):
            L = _lib.load()
            ws = _ksplit_workspace(L.msda_conv_forward_workspace_bytes, (N, H, W, self.Cin, self.Cout, self.KH, self.KW, self.stride, self.pad),
                                   x.device)
            _lib.check(L.msda_conv_forward_ws_bf16(
                x.data_ptr(), self.packed.data_ptr(), self.scale.data_ptr(), self.shift.data_ptr(),
                residual.data_ptr() if residual is not None else None, N, H, W, self.Cin, self.Cout, self.KH, self.KW, self.stride,
                self.pad, int(self.relu), out.data_ptr(), ws.data_ptr() if ws is not None else None, _stream(x.device)))
        return out


def set_tiling(co_tiles=0, pixel_tiles=0):
    """Tuning / tests: force the kernel's per-wave tile (0 = automatic)."""
    _lib.check(_lib.load().msda_conv_set_tiling(int(co_tiles), int(pixel_tiles)))


def set_ring(slots=0):
    """Tuning / tests: the operand rings of the C_in % 64 == 0 convolution kernel: -1 never, 0 automatic, 3 / 4 / 6 slots wherever it applies."""
    _lib.check(_lib.load().msda_conv_set_ring(int(slots)))


def conv_dgrad(dz, packed_t, x_shape, Cout, KH, KW, stride, padding, add=None, relu_out=None):
    """Input gradient of a convolution (``msda_conv_dgrad_fused_bf16``): dz (N, Ho, Wo, Cout) bf16 NHWC, ``packed_t`` the packed flipped /
    transposed weight -> dx of shape ``x_shape`` (N, H, W, Cin); ``add``: a second gradient of the same tensor, summed in the epilogue;
    ``relu_out``: the tensor itself when it is the output of a ReLU -- dx is then the gradient at that ReLU's input"""
    N, H, W, Cin = x_shape
    dx = torch.empty(x_shape, dtype=torch.bfloat16, device=dz.device)
    L = _lib.load()
    with _lib.on_device(dz.device):
        ws = _ksplit_workspace(L.msda_conv_dgrad_workspace_bytes, (N, dz.shape[1], dz.shape[2], Cout, Cin, KH, KW, stride, padding, H, W), dz.device)
        _lib.check(L.msda_conv_dgrad_fused_bf16(dz.data_ptr(), packed_t.data_ptr(), N, dz.shape[1], dz.shape[2], Cout, Cin, KH, KW, stride, padding,
                                                H, W, add.data_ptr() if add is not None else None,
                                                relu_out.data_ptr() if relu_out is not None else None, dx.data_ptr(),
                                                ws.data_ptr() if ws is not None else None, _stream(dz.device)))
    return dx


def conv_forward(x, packed, scale, shift, residual, Cout, KH, KW, stride, padding, relu):
    """``msda_conv_forward_ws_bf16`` on an NHWC bf16 tensor with an already packed weight -> (N, Ho, Wo, Cout) bf16"""
    N, H, W, Cin = x.shape
    Ho, Wo = (H + 2 * padding - KH) // stride + 1, (W + 2 * padding - KW) // stride + 1
    out = torch.empty((N, Ho, Wo, Cout), dtype=torch.bfloat16, device=x.device)
    with _lib.on_device(x.device):
        L = _lib.load()
        ws = _ksplit_workspace(L.msda_conv_forward_workspace_bytes, (N, H, W, Cin, Cout, KH, KW, stride, padding), x.device)
        _lib.check(L.msda_conv_forward_ws_bf16(
            x.data_ptr(), packed.data_ptr(), scale.data_ptr(), shift.data_ptr(), residual.data_ptr() if residual is not None else None,
            N, H, W, Cin, Cout, KH, KW, stride, padding, int(relu), out.data_ptr(), ws.data_ptr() if ws is not None else None,
            _stream(x.device)))
    return out


def conv_wgrad(dz, x, Cout, KH, KW, stride, padding, scale=None):
    """Weight gradient (``msda_conv_wgrad_bf16``): dz (N, Ho, Wo, Cout), x (N, H, W, Cin) bf16 NHWC -> (Cout, Cin, KH, KW) fp32 (nn.Conv2d's
    layout), times ``scale[co]`` when given.  Cout % 128 == 0 and Cin % 128 == 0."""
    N, H, W, Cin = x.shape
    dw = torch.empty((Cout, Cin, KH, KW), dtype=torch.float32, device=x.device)
    L = _lib.load()
    nb = ctypes.c_int64(0)
    _lib.check(L.msda_conv_wgrad_workspace_bytes(N, H, W, Cin, Cout, KH, KW, stride, padding, ctypes.byref(nb)))
    ws = torch.empty(nb.value // 4, dtype=torch.float32, device=x.device) if nb.value else None
    with _lib.on_device(x.device):
        _lib.check(L.msda_conv_wgrad_bf16(dz.data_ptr(), x.data_ptr(), N, H, W, Cin, Cout, KH, KW, stride, padding, dw.data_ptr(), None,
                                          scale.data_ptr() if scale is not None else None, 1, ws.data_ptr() if ws is not None else None,
                                          _stream(x.device)))
    return dw


def conv_wgrad_group(problems):
    """Several weight gradients in one launch (``msda_conv_wgrad_group_bf16``): ``problems`` = [(dz, x, Cout, KH, KW, stride, padding, scale)]
    as :func:`conv_wgrad` takes them (at most 8) -> the list of (Cout, Cin, KH, KW) fp32 results"""
    L = _lib.load()
    n = len(problems)
    arr = (_lib.WgradProblem * n)()
    outs = []
    dev = problems[0][0].device
    for j, (dz, x, Cout, KH, KW, stride, padding, scale) in enumerate(problems):
        N, H, W, Cin = x.shape
        dw = torch.empty((Cout, Cin, KH, KW), dtype=torch.float32, device=dev)
        outs.append(dw)
        arr[j] = _lib.WgradProblem(dz.data_ptr(), x.data_ptr(), dw.data_ptr(), scale.data_ptr() if scale is not None else None, None,
                                   N, H, W, Cin, Cout, KH, KW, stride, padding)
    nb = ctypes.c_int64(0)
    _lib.check(L.msda_conv_wgrad_group_workspace_bytes(arr, n, ctypes.byref(nb)))
    ws = torch.empty(nb.value // 4, dtype=torch.float32, device=dev) if nb.value else None
    with _lib.on_device(dev):
        _lib.check(L.msda_conv_wgrad_group_bf16(arr, n, ws.data_ptr() if ws is not None else None, _stream(dev)))
    return outs


def _pool(x, k, stride, pad, is_max):
    assert x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[3] % 8 == 0
    x = x.contiguous()
    N, H, W, C = x.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = torch.empty((N, Ho, Wo, C), dtype=torch.bfloat16, device=x.device)
    with _lib.on_device(x.device):
        _lib.check(_lib.load().msda_pool_nhwc_bf16(x.data_ptr(), N, H, W, C, k, stride, pad, int(is_max), out.data_ptr(), _stream(x.device)))
    return out


def avg_pool_nhwc(x, k):
    """nn.AvgPool2d(k) on an NHWC bf16 tensor (clip/model.py:24, :36, :115): ``msda_pool_nhwc_bf16`` (fp32 sums, one rounding)"""
    return x if k == 1 else _pool(x, k, k, 0, False)


def max_pool_nhwc(x, k=3, stride=2, pad=1):
    """nn.MaxPool2d(k, stride, pad) on an NHWC bf16 tensor (torchvision's ResNet stem)"""
    return _pool(x, k, stride, pad, True)


def group_norm8_nhwc(x, gamma, beta, eps=1e-5, want_f32=True, want_bf16=True):
    """nn.GroupNorm(C // 8, C) on an NHWC bf16 tensor (N, H, W, C): -> (fp32 result or None, bf16 result or None), both (N, H, W, C)"""
    assert x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[3] % 8 == 0
    x = x.contiguous()
    N, H, W, C = x.shape
    stats = torch.empty(N * (C // 8) * 2, dtype=torch.float64, device=x.device)
    o32 = torch.empty((N, H, W, C), dtype=torch.float32, device=x.device) if want_f32 else None
    o16 = torch.empty((N, H, W, C), dtype=torch.bfloat16, device=x.device) if want_bf16 else None
    g, b = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
    with _lib.on_device(x.device):
        _lib.check(_lib.load().msda_groupnorm8_nhwc_bf16(x.data_ptr(), g.data_ptr(), b.data_ptr(), float(eps), N, H * W, C, stats.data_ptr(),
                                                         o32.data_ptr() if o32 is not None else None,
                                                         o16.data_ptr() if o16 is not None else None, _stream(x.device)))
    return o32, o16


class GroupNorm8Function(torch.autograd.Function):
    """nn.GroupNorm(C // 8, C) on an NHWC bf16 tensor with its gradients on the library's kernels (``msda_groupnorm8_nhwc_bf16`` /
    ``msda_groupnorm8_backward_nhwc_bf16``): bf16 in, bf16 out, fp32 / fp64 statistics; ``apply(x (N, H, W, C), gamma, beta, eps)``"""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        assert x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[3] % 8 == 0
        x = x.contiguous()
        N, H, W, C = x.shape
        stats = torch.empty(N * (C // 8) * 2, dtype=torch.float64, device=x.device)
        out = torch.empty_like(x)
        g, b = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        with _lib.on_device(x.device):
            _lib.check(_lib.load().msda_groupnorm8_nhwc_bf16(x.data_ptr(), g.data_ptr(), b.data_ptr(), float(eps), N, H * W, C, stats.data_ptr(),
                                                             None, out.data_ptr(), _stream(x.device)))
        ctx.save_for_backward(x, g, stats)
        ctx.eps, ctx.dts = float(eps), (gamma.dtype, beta.dtype)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, g, stats = ctx.saved_tensors
        N, H, W, C = x.shape
        dy = dy.to(torch.bfloat16).contiguous()
        dx = torch.empty_like(x)
        dgb = torch.empty(2, C, dtype=torch.float32, device=x.device)
        bstats = torch.empty(N * (C // 8) * 16, dtype=torch.float64, device=x.device)
        with _lib.on_device(x.device):
            _lib.check(_lib.load().msda_groupnorm8_backward_nhwc_bf16(x.data_ptr(), dy.data_ptr(), g.data_ptr(), ctx.eps, N, H * W, C,
                                                                      stats.data_ptr(), bstats.data_ptr(), dx.data_ptr(), dgb[0].data_ptr(),
                                                                      dgb[1].data_ptr(), _stream(x.device)))
        return dx, dgb[0].to(ctx.dts[0]), dgb[1].to(ctx.dts[1]), None


# ---- training: the same convolution with gradients ----------------------------------------------------------------------------------
def _pack(w):
    L = _lib.load()
    Cout, Cin, KH, KW = w.shape
    n = ctypes.c_int64(0)
    _lib.check(L.msda_conv_packed_elems(Cout, Cin, KH, KW, ctypes.byref(n)))
    packed = torch.empty(n.value, dtype=torch.int16, device=w.device)
    with _lib.on_device(w.device):
        _lib.check(L.msda_conv_pack_weight(w.data_ptr(), Cout, Cin, KH, KW, packed.data_ptr(), _stream(w.device)))
    return packed


class PackCache:
    """Packed forms of ONE convolution weight, owned by the module that owns the parameter (no global table: a freed tensor's address
    can come back with the same version counter): the forward weight and the flipped / transposed / scale-folded weight of the input
    gradient, refreshed when the parameter (or the scale) has been modified in place -- i.e. after an optimizer step, a
    ``load_state_dict`` or any other in-place op that autograd's version counter sees.  Writes through ``param.data`` bypass that
    counter: call :meth:`clear` after them."""

    def __init__(self):
        self._slots = {}

    def clear(self):
        self._slots.clear()

    def get(self, weight, scale, transposed):
        ver = (weight.data_ptr(), weight._version, scale.data_ptr() if transposed else 0, scale._version if transposed else 0,
               tuple(weight.shape))
        hit = self._slots.get(transposed)
        if hit is not None and hit[0] == ver:
            return hit[1]
        packed = _pack_form(weight, scale, transposed)
        self._slots[transposed] = (ver, packed)
        return packed


def _pack_form(weight, scale, transposed):
    w = weight.detach().float()
    if transposed:
        w = (w * scale.view(-1, 1, 1, 1)).flip(2, 3).permute(1, 0, 2, 3)          # (Cin, Cout, KH, KW)
    return _pack(w.contiguous())


def _packed_for(weight, scale, transposed, cache=None):
    return cache.get(weight, scale, transposed) if cache is not None else _pack_form(weight, scale, transposed)


class ConvAffineFunction(torch.autograd.Function):
    """y = act(scale * conv(x, weight) + shift (+ residual)) on NHWC bf16 activations with gradients for x, weight and residual
    (scale is the frozen BatchNorm's: no gradient, backbone.py:20-56; shift gets one when it is a trained bias).  Forward: ``msda_conv_forward_bf16``.  Backward: the ReLU
    mask is a PyTorch element-wise op; the input gradient is ``msda_conv_dgrad_bf16`` (the forward kernel on the zero-upsampled output
    gradient with the flipped, transposed, scale-folded weight); the weight gradient is ``msda_conv_wgrad_bf16`` (csrc/conv_wgrad.hip:
    pixel-contraction GEMM per tap with transposed LDS reads) when both channel counts are multiples of 128 -- every convolution of
    ResNet-50's trained stages -- and MIOpen's (``aten::convolution_backward``) otherwise."""

    library_wgrad = False      # tests / timing: True sends every weight gradient to MIOpen

    @staticmethod
    def forward(ctx, x, weight, scale, shift, residual, stride, padding, relu, cache=None):
        assert x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4
        Cout, Cin, KH, KW = weight.shape
        x = x.contiguous()
        N, H, W, _ = x.shape
        Ho, Wo = (H + 2 * padding - KH) // stride + 1, (W + 2 * padding - KW) // stride + 1
        out = torch.empty((N, Ho, Wo, Cout), dtype=torch.bfloat16, device=x.device)
        res = residual.contiguous() if residual is not None else None
        packed = _packed_for(weight, scale, False, cache)
        with _lib.on_device(x.device):
            L = _lib.load()
            ws = _ksplit_workspace(L.msda_conv_forward_workspace_bytes, (N, H, W, Cin, Cout, KH, KW, stride, padding), x.device)
            _lib.check(L.msda_conv_forward_ws_bf16(
                x.data_ptr(), packed.data_ptr(), scale.data_ptr(), shift.data_ptr(), res.data_ptr() if res is not None else None,
                N, H, W, Cin, Cout, KH, KW, stride, padding, int(relu), out.data_ptr(), ws.data_ptr() if ws is not None else None,
                _stream(x.device)))
        ctx.save_for_backward(x, weight, scale, out if relu else None)
        ctx.cfg = (stride, padding, relu, residual is not None, cache)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, weight, scale, out = ctx.saved_tensors
        stride, padding, relu, has_res, cache = ctx.cfg
        Cout, Cin, KH, KW = weight.shape
        N, H, W, _ = x.shape
        dz = dy.contiguous()
        if relu:
            dz = torch.ops.aten.threshold_backward(dz, out, 0)     # gradient at the ReLU's input (= the residual's gradient)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            if Cout % 32 or Cin % 16 or KH != KW:
                raise RuntimeError("ConvAffineFunction: the input gradient needs C_out % 32 == 0, C_in % 16 == 0 and a square kernel")
            packed_t = _packed_for(weight, scale, True, cache)
            dx = torch.empty_like(x)
            with _lib.on_device(x.device):
                L = _lib.load()
                ws = _ksplit_workspace(L.msda_conv_dgrad_workspace_bytes, (N, dz.shape[1], dz.shape[2], Cout, Cin, KH, KW, stride, padding, H, W),
                                       x.device)
                _lib.check(L.msda_conv_dgrad_ws_bf16(dz.data_ptr(), packed_t.data_ptr(), N, dz.shape[1], dz.shape[2], Cout, Cin, KH, KW, stride,
                                                     padding, H, W, dx.data_ptr(), ws.data_ptr() if ws is not None else None, _stream(x.device)))
        if ctx.needs_input_grad[1]:
            # the weight gradient is taken with dz and scaled per output channel afterwards (the affine's scale commutes with the sum)
            if Cout % 128 == 0 and Cin % 128 == 0 and not ConvAffineFunction.library_wgrad:
                dw = torch.empty((Cout, Cin, KH, KW), dtype=torch.float32, device=x.device)      # nn.Conv2d's layout, scaled: final
                L = _lib.load()
                nb = ctypes.c_int64(0)
                _lib.check(L.msda_conv_wgrad_workspace_bytes(N, H, W, Cin, Cout, KH, KW, stride, padding, ctypes.byref(nb)))
                ws = torch.empty(nb.value // 4, dtype=torch.float32, device=x.device) if nb.value else None
                with _lib.on_device(x.device):
                    _lib.check(L.msda_conv_wgrad_bf16(dz.data_ptr(), x.data_ptr(), N, H, W, Cin, Cout, KH, KW, stride, padding,
                                                      dw.data_ptr(), None, scale.data_ptr(), 1, ws.data_ptr() if ws is not None else None,
                                                      _stream(x.device)))
                dw = dw.to(weight.dtype)
            else:       # channel counts the wgrad kernel does not take (ResNet-50's layer2-4 never get here): MIOpen
                _, dw, _ = torch.ops.aten.convolution_backward(dz.permute(0, 3, 1, 2), x.permute(0, 3, 1, 2),
                                                               weight.detach().to(torch.bfloat16), None, [stride, stride],
                                                               [padding, padding], [1, 1], False, [0, 0], 1, [False, True, False])
                dw = (dw.float() * scale.view(-1, 1, 1, 1)).to(weight.dtype)
        dshift = dz.float().sum(dim=(0, 1, 2)) if ctx.needs_input_grad[3] else None      # a trained bias (input projections)
        return dx, dw, None, dshift, (dz if has_res else None), None, None, None, None


class ConvBNAct(torch.nn.Module):
    """nn.Conv2d(bias = False) + FrozenBatchNorm2d (+ residual) (+ ReLU) as one trainable module on NHWC bf16 activations: ``weight`` is
    the fp32 parameter under the name nn.Conv2d gives it; the four FrozenBatchNorm2d tensors are buffers (backbone.py:28-33)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, relu=True):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size))
        torch.nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        self.register_buffer("bn_weight", torch.ones(out_channels))
        self.register_buffer("bn_bias", torch.zeros(out_channels))
        self.register_buffer("running_mean", torch.zeros(out_channels))
        self.register_buffer("running_var", torch.ones(out_channels))
        self.stride, self.padding, self.relu = stride, padding, relu
        self._pack_cache = PackCache()

    def scale_shift(self):
        """the frozen affine, folded once and again only after the buffers changed (a fresh tensor per call would defeat the
        PackCache's key, or -- at a recycled address -- alias a stale scale-folded weight)"""
        ver = (self.bn_weight._version, self.bn_bias._version, self.running_mean._version, self.running_var._version,
               self.bn_weight.data_ptr())
        if getattr(self, "_folded_ver", None) != ver:
            self._folded = fold_bn(self.bn_weight, self.bn_bias, self.running_mean, self.running_var, 1e-5)
            self._folded_ver = ver
        return self._folded

    def forward(self, x, residual=None):
        scale, shift = self.scale_shift()
        return ConvAffineFunction.apply(x, self.weight, scale, shift, residual, self.stride, self.padding, self.relu, self._pack_cache)
