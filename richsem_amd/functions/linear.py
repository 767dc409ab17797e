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
    with _lib.on_device(x.device):
        _lib.check(L.msda_conv_wgrad_bf16(dy.data_ptr(), x.data_ptr(), 1, 1, T, cin, cout, 1, 1, 1, 0, dw.data_ptr(),
                                          db.data_ptr() if db is not None else None, None, 0, ws.data_ptr() if ws is not None else None,
                                          _lib.raw_stream(x.device)))
    return (dw, db) if with_bias else dw


class LinearBf16Function(torch.autograd.Function):
    """``F.linear`` on bf16 activations with fp32 parameters (cast per call): the forward and the input gradient are the library's bf16
    GEMMs, the weight gradient is :func:`linear_wgrad_bf16` (for layer sizes it supports and enough tokens to pay: the library's
    transposed GEMM otherwise), the bias gradient a column sum.  Gradients come back in the parameters' dtype."""

    MIN_TOKENS = 1024

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
    with _lib.on_device(w.device):
        _lib.check(_lib.load().msda_lin256_pack_bf16(w.data_ptr(), w.shape[0], 256, packed.data_ptr(),
                                                     _lib.raw_stream(w.device)))
    return packed


def lin256(x, packed_w, bias=None, relu=False, relu_mask=None, row_mask=None):
    """x (T, 256) bf16 -> (T, out_features) bf16: ``x W^T + bias`` (``relu``: with ReLU), or ``(x W^T) * (relu_mask > 0)`` when
    ``relu_mask`` (T, out_features) bf16 is given (the gradient at a ReLU's input from the gradient at its output), or
    ``x W^T + bias`` with the rows of masked tokens zeroed when ``row_mask`` (T,) bool is given (MSDeformAttn's value projection
    with its padding mask, ops/modules/ms_deform_attn.py:94-96)"""
    assert x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 2 and x.shape[1] == 256
    x = x.contiguous()
    N = packed_w.shape[0]
    out = torch.empty((x.shape[0], N), dtype=torch.bfloat16, device=x.device)
    if relu_mask is not None:
        assert relu_mask.shape == out.shape and relu_mask.dtype == torch.bfloat16 and relu_mask.is_contiguous() and bias is None
    aux = relu_mask
    epi = 2 if relu_mask is not None else (1 if relu else 0)
    if row_mask is not None:
        assert relu_mask is None and not relu and row_mask.dtype == torch.bool and row_mask.numel() == x.shape[0]
        aux, epi = row_mask.contiguous().view(torch.uint8), 3
    b = bias if bias is None or (bias.dtype == torch.float32 and bias.is_contiguous() and not bias.requires_grad) \
        else bias.detach().float().contiguous()
    with _lib.on_device(x.device):
        _lib.check(_lib.load().msda_lin256_forward_bf16(x.data_ptr(), packed_w.data_ptr(), b.data_ptr() if b is not None else None,
                                                        aux.data_ptr() if aux is not None else None, epi, x.shape[0], 256, N,
                                                        out.data_ptr(), _lib.raw_stream(x.device)))
    return out


def lin256_f32_pack(weight):
    """weight (out_features, 256) float32 -> bf16 hi + lo parts in the fragment order of ``lin256_f32``; out_features % 32 == 0"""
    w = weight.detach().float().contiguous()
    assert w.is_cuda and w.dim() == 2 and w.shape[1] == 256 and w.shape[0] % 32 == 0
    packed = torch.empty(2 * w.numel(), dtype=torch.int16, device=w.device)
    with _lib.on_device(w.device):
        _lib.check(_lib.load().msda_lin256_pack_f32(w.data_ptr(), w.shape[0], 256, packed.data_ptr(),
                                                    _lib.raw_stream(w.device)))
    return packed


def lin256_f32(x, packed_w, out_features, bias=None):
    """x (T, 256) float32 -> x W^T + bias (T, out_features) float32 at fp32-level accuracy on bf16 MFMAs (three per tile)"""
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] == 256
    x = x.contiguous()
    out = torch.empty((x.shape[0], out_features), dtype=torch.float32, device=x.device)
    b = bias.detach().float().contiguous() if bias is not None else None
    with _lib.on_device(x.device):
        _lib.check(_lib.load().msda_lin256_forward_f32(x.data_ptr(), packed_w.data_ptr(), b.data_ptr() if b is not None else None,
                                                       x.shape[0], 256, out_features, out.data_ptr(),
                                                       _lib.raw_stream(x.device)))
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


class VersionCache:
    """A derived form of a few parameters (bf16 casts, packed weights, stacked projections) kept across calls and rebuilt when one of
    them has been modified in place (optimizer step, ``load_state_dict``, ``copy_``: whatever bumps autograd's version counter).  Writes
    THROUGH ``param.data`` bypass that counter: call :meth:`clear` after them."""

    def __init__(self):
        self._ver = self._val = None

    def clear(self):
        self._ver = self._val = None

    def get(self, params, build):
        ver = tuple((p.data_ptr(), p._version) for p in params)
        if ver != self._ver:
            with torch.no_grad():
                self._val = build()
            self._ver = ver
        return self._val


def pack_linear256(weights, biases):
    """The forms :class:`Lin256Function` runs a (stack of) ``nn.Linear(256, n_i)`` from: the weights concatenated along the output
    dimension as bf16 (``w16``, for the input gradient), in lin256's fragment order (``packed``), the transposed weight packed the
    same way when the stack is 256 wide (``packed_t``: the input gradient is then a K = 256 product too), and the float32 bias."""
    w = torch.cat([t.detach() for t in weights], 0) if len(weights) > 1 else weights[0].detach()
    b = torch.cat([t.detach() for t in biases], 0) if len(biases) > 1 else biases[0].detach()
    w16 = w.to(torch.bfloat16).contiguous()
    n = w16.shape[0]
    assert w16.shape[1] == 256 and n % 64 == 0
    return {"w16": w16, "packed": lin256_pack(w16), "packed_t": lin256_pack(w16.t().contiguous()) if n == 256 else None,
            "b32": b.float().contiguous(), "rows": [int(t.shape[0]) for t in weights]}


def pack_linear256_padded(weight, bias):
    """:func:`pack_linear256` for a layer with fewer than 64 outputs (the box heads' 256 -> 4): the weight padded with zero rows to 64;
    the caller takes the first ``out_features`` columns of the result"""
    n = weight.shape[0]
    w = torch.zeros((64, 256), dtype=weight.dtype, device=weight.device)
    b = torch.zeros(64, dtype=bias.dtype, device=bias.device)
    w[:n], b[:n] = weight.detach(), bias.detach()
    pk = pack_linear256([w], [b])
    pk["rows"] = [n]
    return pk


class Lin256NarrowFunction(torch.autograd.Function):
    """``x W^T + b`` for a 256 -> n layer with n <= 8 (the box heads' 256 -> 4): forward on csrc/lin256_mfma.hip with the weight padded to
    64 rows (``pk`` from :func:`pack_linear256_padded`), of which the first n columns are handed out; backward as two streaming kernels
    (``msda_narrow_linear_backward_bf16``) -- not the three GEMMs with a zero-padded 64-wide gradient that autograd's slice would feed.
    ``apply(x, pk, weight, bias)`` -> (..., n) bf16"""

    @staticmethod
    def forward(ctx, x, pk, weight, bias):
        x2 = x.reshape(-1, 256)
        n = weight.shape[0]
        out = lin256(x2, pk["packed"], pk["b32"])[:, :n].contiguous()
        ctx.save_for_backward(x2, weight)
        ctx.shape, ctx.dts = x.shape, (weight.dtype, bias.dtype)
        return out.view(x.shape[:-1] + (n,))

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x2, weight = ctx.saved_tensors
        n, T = weight.shape[0], x2.shape[0]
        dy2 = dy.reshape(T, n).to(torch.bfloat16).contiguous()
        x2 = x2.contiguous()
        w32 = weight.detach().float().contiguous()
        dx = torch.empty((T, 256), dtype=torch.bfloat16, device=x2.device) if ctx.needs_input_grad[0] else None
        dwb = torch.empty(n * 256 + 8, dtype=torch.float32, device=x2.device)
        with _lib.on_device(x2.device):
            _lib.check(_lib.load().msda_narrow_linear_backward_bf16(
                dy2.data_ptr(), x2.data_ptr(), w32.data_ptr(), T, n, dx.data_ptr() if dx is not None else None, dwb.data_ptr(),
                dwb[n * 256:].data_ptr(), _lib.raw_stream(x2.device)))
        dw, db = dwb[:n * 256].view(n, 256), dwb[n * 256:n * 256 + n]
        return (dx.view(ctx.shape) if dx is not None else None, None, dw.to(ctx.dts[0]) if ctx.needs_input_grad[2] else None,
                db.to(ctx.dts[1]) if ctx.needs_input_grad[3] else None)


def _mask_rows_(t, mask):
    """zero the rows of ``t`` (T, C) bf16 where ``mask`` (T,) bool is set, in place (only those rows are touched)"""
    with _lib.on_device(t.device):
        _lib.check(_lib.load().msda_mask_rows_bf16(t.data_ptr(), mask.view(torch.uint8).data_ptr(), mask.numel(), t.numel() // mask.numel(),
                                                   _lib.raw_stream(t.device)))
    return t


# ---- a layer's weight gradients in ONE launch ----------------------------------------------------------------------------------------
class WgradGroup:
    """The weight (+ bias) gradients of one layer call, deferred and launched together (``msda_conv_wgrad_group_bf16``): at the decoder's
    ~2 k tokens every such product is a launch of 11-18 us plus a 6 us reduction that keep the matrix pipe 1-4 % busy; seven of them per
    decoder layer in one launch share the chip.  Protocol (modules/decoder_layer.py): the layer routes its parameters through
    :func:`wgrad_boundary` -- an identity whose backward runs AFTER every function that consumed one of its outputs -- and runs its body
    ``with group:``; :class:`Lin256Function` / ``FFNSmallFunction`` then hand autograd EMPTY gradient tensors and register the products here;
    the boundary's backward launches them and passes the (now filled) tensors on to the parameters.

    The protocol is ENFORCED, not assumed (an unwritten gradient would otherwise be read without any error):
      * a function defers only when every parameter it was given is an alias made by THIS group's boundary (:meth:`owns`) -- called on raw
        module parameters inside an active group it computes its gradients on the spot;
      * the boundary's backward checks that every tensor it filled is (a view of) one of the gradients autograd handed it -- an alias that
        fed two functions arrives as a SUM, a new tensor formed before the flush: that raises instead of training on garbage."""

    enabled = True
    _active = None

    def __init__(self):
        self.pending = []
        self._aliases = ()      # weak references (an alias -> its grad_fn -> the boundary's ctx -> this group: a strong one would be a cycle through C++)

    def __enter__(self):
        self._prev, WgradGroup._active = WgradGroup._active, self
        return self

    def __exit__(self, *exc):
        WgradGroup._active = self._prev
        return False

    @staticmethod
    def active():
        return WgradGroup._active

    @staticmethod
    def active_for(params):
        """the active group if every tensor of ``params`` is one of its boundary's aliases, else None (no deferral)"""
        g = WgradGroup._active
        return g if g is not None and g.owns(params) else None

    def owns(self, params):
        mine = {id(a) for a in (r() for r in self._aliases) if a is not None}
        return bool(params) and all(id(p) in mine for p in params)

    def add(self, dy2, x2, with_bias):
        """register dW = dy2^T x2 (+ db = column sums of dy2): -> (dw (out, in) float32, db (out) float32 or None), written by flush()"""
        dw = torch.empty((dy2.shape[1], x2.shape[1]), dtype=torch.float32, device=x2.device)
        db = torch.empty(dy2.shape[1], dtype=torch.float32, device=x2.device) if with_bias else None
        self.pending.append((dy2, x2, dw, db))
        return dw, db

    def flush(self):
        """launch the registered products; -> the tensors that were filled (for :meth:`verify`)"""
        pend, self.pending = self.pending, []
        self._launch(pend)
        return [t for (_, _, dw, db) in pend for t in (dw, db) if t is not None]

    @staticmethod
    def verify(filled, grads):
        """every tensor the flush wrote must be (a view of) a gradient autograd routed through the boundary"""
        seen = {g.untyped_storage().data_ptr() for g in grads if g is not None}
        for t in filled:
            if t.untyped_storage().data_ptr() not in seen:
                raise RuntimeError(
                    "WgradGroup: a deferred weight gradient did not arrive at its boundary as the tensor that was registered -- a parameter "
                    "alias fed more than one function (autograd summed an unwritten tensor), or its gradient was cast / hooked on the way. "
                    "Every alias of wgrad_boundary() must feed exactly one deferring function (richsem_amd/functions/linear.py)")

    @staticmethod
    def _launch(pend):
        L = _lib.load()
        for i0 in range(0, len(pend), 8):
            part = pend[i0:i0 + 8]
            arr = (_lib.WgradProblem * len(part))()
            for j, (dy2, x2, dw, db) in enumerate(part):
                arr[j] = _lib.WgradProblem(dy2.data_ptr(), x2.data_ptr(), dw.data_ptr(), None, db.data_ptr() if db is not None else None,
                                           1, 1, x2.shape[0], x2.shape[1], dy2.shape[1], 1, 1, 1, 0)
            nb = ctypes.c_int64(0)
            _lib.check(L.msda_conv_wgrad_group_workspace_bytes(arr, len(part), ctypes.byref(nb)))
            dev = part[0][1].device
            ws = torch.empty(nb.value // 4, dtype=torch.float32, device=dev) if nb.value else None
            with _lib.on_device(dev):
                _lib.check(L.msda_conv_wgrad_group_bf16(arr, len(part), ws.data_ptr() if ws is not None else None, _lib.raw_stream(dev)))


class WgradBoundary(torch.autograd.Function):
    """``apply(group, *params)`` -> aliases of ``params``; the backward first launches ``group``'s deferred weight gradients (every
    function downstream of the aliases has run its backward by then), then passes the aliases' gradients through to the parameters"""

    @staticmethod
    def forward(ctx, group, *params):
        ctx.group = group
        ctx.set_materialize_grads(False)      # (an alias that fed nothing comes back as None, not as a zero tensor)
        return tuple(p.view_as(p) for p in params)

    @staticmethod
    def backward(ctx, *grads):
        WgradGroup.verify(ctx.group.flush(), grads)
        return (None,) + grads


def wgrad_boundary(group, *params):
    """aliases of ``params`` behind ``group``'s :class:`WgradBoundary`; only these aliases may be deferred to the group"""
    import weakref
    aliases = WgradBoundary.apply(group, *params)
    group._aliases = tuple(weakref.ref(a) for a in aliases)
    return aliases


def deferrable(group, dy2, x2, dts):
    """``group`` (the :class:`WgradGroup` that was active in the forward, or None) if the product dy2^T x2 can be deferred to it: the
    weight-gradient kernel's shapes, enough tokens, float32 parameters"""
    if group is None or not linear_wgrad_supported(dy2.shape[1], x2.shape[1]) or dy2.shape[0] < LinearBf16Function.MIN_TOKENS:
        return None
    return group if all(dt == torch.float32 for dt in dts) else None


class Lin256Function(torch.autograd.Function):
    """``x W^T + b`` for a 256-wide bf16 input on the library's own MFMA kernel (csrc/lin256_mfma.hip): forward, and the input gradient
    too when the layer is 256 -> 256 (else the library's bf16 GEMM); the weight / bias gradients on the weight-gradient kernel where
    there are enough tokens to pay (functions/linear.py: linear_wgrad_bf16), else the library's transposed GEMM.
    ``apply(x, pk, row_mask, relu, *params)``: ``pk`` from :func:`pack_linear256` (kept by the caller in a :class:`VersionCache`);
    ``row_mask`` (tokens,) bool or None zeroes the rows of masked tokens in the kernel's epilogue (and their gradient); ``relu``: the
    ReLU in the epilogue as well; ``params`` = the weights then the biases of the stacked layers, to which the gradients are routed."""

    @staticmethod
    def forward(ctx, x, pk, row_mask, relu, *params):
        x2 = x.reshape(-1, 256)
        out = lin256(x2, pk["packed"], pk["b32"], relu=bool(relu), row_mask=row_mask.reshape(-1) if row_mask is not None else None)
        ctx.save_for_backward(x, row_mask, out if relu else None)
        ctx.pk, ctx.dts = pk, tuple(p.dtype for p in params)
        ctx.group = WgradGroup.active_for(params)
        return out.view(x.shape[:-1] + (out.shape[-1],))

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, row_mask, out = ctx.saved_tensors
        pk, dts = ctx.pk, ctx.dts
        n = pk["w16"].shape[0]
        dy2 = dy.reshape(-1, n)
        if out is not None:           # gradient at the ReLU's input
            dy2 = torch.ops.aten.threshold_backward(dy2.contiguous(), out, 0)
        if row_mask is not None:      # (a copy: the incoming gradient is not ours to modify)
            dy2 = _mask_rows_(dy2.clone(memory_format=torch.contiguous_format), row_mask.reshape(-1))
        elif not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        x2 = x.reshape(-1, 256)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = (lin256(dy2, pk["packed_t"]) if pk["packed_t"] is not None else dy2 @ pk["w16"]).view(x.shape)
        need = ctx.needs_input_grad[4:]
        nl = len(pk["rows"])
        grads = [None] * (2 * nl)
        if any(need):
            want_w, want_b = any(need[:nl]), any(need[nl:])
            dw = db = None
            grp = deferrable(ctx.group, dy2, x2, dts) if want_w else None
            if grp is not None:      # registered with the layer's group: written when its WgradBoundary runs
                dw, db = grp.add(dy2, x2.contiguous(), want_b)
            elif want_w and linear_wgrad_supported(n, 256) and dy2.shape[0] >= LinearBf16Function.MIN_TOKENS:
                if want_b:
                    dw, db = linear_wgrad_bf16(dy2, x2.contiguous(), with_bias=True)
                else:
                    dw = linear_wgrad_bf16(dy2, x2.contiguous())
            elif want_w:
                dw = (dy2.t() @ x2).float()
            if want_b and db is None:
                db = dy2.sum(0, dtype=torch.float32)
            r0 = 0
            for i, r in enumerate(pk["rows"]):
                if need[i] and dw is not None:
                    grads[i] = dw[r0:r0 + r].to(dts[i])
                if need[nl + i] and db is not None:
                    grads[nl + i] = db[r0:r0 + r].to(dts[nl + i])
                r0 += r
        return (dx, None, None, None) + tuple(grads)


class StackedValueProjFunction(torch.autograd.Function):
    """Several 256 -> 256 layers applied to ONE input in one product (``msda_lin256_forward_stacked_bf16``): the cross-attention value
    projections of all decoder layers (modules/decoder.py).  ``apply(x, pk, row_mask, *weights, *biases)`` -> a tuple of contiguous
    (..., 256) tensors, one per layer; the backward stacks the layers' gradients once and runs ONE input-gradient product (K = 256 x
    layers) and ONE weight-gradient product."""

    @staticmethod
    def forward(ctx, x, pk, row_mask, *params):
        x2 = x.reshape(-1, 256).contiguous()
        nl = len(pk["rows"])
        out = torch.empty((nl,) + x.shape[:-1] + (256,), dtype=torch.bfloat16, device=x.device)
        m8 = row_mask.reshape(-1).contiguous().view(torch.uint8) if row_mask is not None else None
        with _lib.on_device(x.device):
            _lib.check(_lib.load().msda_lin256_forward_stacked_bf16(
                x2.data_ptr(), pk["packed"].data_ptr(), pk["b32"].data_ptr(), m8.data_ptr() if m8 is not None else None, x2.shape[0], 256,
                256 * nl, out.data_ptr(), _lib.raw_stream(x.device)))
        ctx.save_for_backward(x2, row_mask)
        ctx.pk, ctx.dts, ctx.shape = pk, tuple(p.dtype for p in params), x.shape
        return tuple(out.unbind(0))

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *dys):
        x2, row_mask = ctx.saved_tensors
        pk, dts = ctx.pk, ctx.dts
        nl = len(pk["rows"])
        T = x2.shape[0]
        zero = None
        cols = []
        for dy in dys:       # (a layer whose output took no part in the loss hands over None)
            if dy is None:
                zero = torch.zeros((T, 256), dtype=torch.bfloat16, device=x2.device) if zero is None else zero
                cols.append(zero)
            else:
                cols.append(dy.reshape(T, 256))
        dy_all = torch.cat(cols, 1)                                   # (T, 256 * layers): one copy, then two products
        if row_mask is not None:
            _mask_rows_(dy_all, row_mask.reshape(-1))
        dx = (dy_all @ pk["w16"]).view(ctx.shape) if ctx.needs_input_grad[0] else None
        need = ctx.needs_input_grad[3:]
        grads = [None] * (2 * nl)
        if any(need):
            if linear_wgrad_supported(256 * nl, 256) and T >= LinearBf16Function.MIN_TOKENS:
                dw, db = linear_wgrad_bf16(dy_all, x2, with_bias=True)
            else:
                dw, db = (dy_all.t() @ x2).float(), dy_all.sum(0, dtype=torch.float32)
            for i in range(nl):
                if need[i]:
                    grads[i] = dw[256 * i:256 * (i + 1)].to(dts[i])
                if need[nl + i]:
                    grads[nl + i] = db[256 * i:256 * (i + 1)].to(dts[nl + i])
        return (dx, None, None) + tuple(grads)
