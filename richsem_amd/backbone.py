"""The detector backbone's forward on the MFMA convolution kernel: ResNet-50 with FrozenBatchNorm2d
(models/richsem/backbone.py:20-56, :123-158: torchvision's resnet50 with ``norm_layer = FrozenBatchNorm2d``, stages layer2..layer4
returned for the 4-scale configuration) and the input projections models/richsem/richsem.py:295-310, :593-612 (1 x 1 convolution +
GroupNorm(32) per stage, 3 x 3 stride-2 convolution + GroupNorm for the extra level).

torchvision is a third-party dependency outside the reference tree (``torchvision>=0.6.0``, unpinned) and absent from the image; the
architecture restated here is its published ResNet-50 v1.5 (stem 7 x 7 stride 2 + 3 x 3 stride-2 max pool; bottlenecks [3, 4, 6, 3] with
the stride on the 3 x 3 convolution; projection shortcut = 1 x 1 stride-s convolution + norm), read from the parameter names of its
``state_dict`` (conv1, bn1, layer{1..4}.{i}.conv{1,2,3} / bn{1,2,3} / downsample.{0,1}).  The frozen affine, the ReLU and the residual
add run in the convolution's epilogue, activations are NHWC bf16.  ``ResNet50Frozen`` is the inference form (weights packed once);
``ResNet50`` below is the trainable ``nn.Module`` (forward, input gradient and weight gradient of layer2-4 on the library's kernels).
"""
import torch
import torch.nn.functional as F

from .conv import ConvAffine, fold_bn, group_norm8_nhwc, max_pool_nhwc, to_nhwc_bf16


def _conv_bn(sd, conv, bn, dev, stride=1, padding=0, relu=True):
    scale, shift = fold_bn(sd[bn + ".weight"].to(dev), sd[bn + ".bias"].to(dev), sd[bn + ".running_mean"].to(dev),
                           sd[bn + ".running_var"].to(dev), 1e-5)                     # backbone.py:51-55
    return ConvAffine(sd[conv + ".weight"].to(dev), scale, shift, stride, padding, relu)


class _Bottleneck:
    def __init__(self, sd, p, stride, dev):
        self.conv1 = _conv_bn(sd, p + "conv1", p + "bn1", dev)
        self.conv2 = _conv_bn(sd, p + "conv2", p + "bn2", dev, stride=stride, padding=1)
        self.conv3 = _conv_bn(sd, p + "conv3", p + "bn3", dev, relu=True)              # relu after the residual add: fused
        self.down = _conv_bn(sd, p + "downsample.0", p + "downsample.1", dev, stride=stride, relu=False) \
            if p + "downsample.0.weight" in sd else None

    def __call__(self, x):
        identity = x if self.down is None else self.down(x)
        return self.conv3(self.conv2(self.conv1(x)), residual=identity)


class ResNet50Frozen:
    """``state_dict``: torchvision resnet's (any depth with bottleneck blocks).  ``__call__(images)`` returns the NHWC bf16 outputs of
    the stages named in ``return_layers`` (default layer2, layer3, layer4 = backbone.py's return_interm_indices [1, 2, 3])."""

    def __init__(self, state_dict, return_layers=(2, 3, 4), device="cuda"):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("Not implemented on the CPU")
        sd = {k: v.detach() for k, v in state_dict.items()}
        self.stem = _conv_bn(sd, "conv1", "bn1", dev, stride=2, padding=3)
        self.layers = []
        for li in range(1, 5):
            blocks, b = [], 0
            while f"layer{li}.{b}.conv1.weight" in sd:
                blocks.append(_Bottleneck(sd, f"layer{li}.{b}.", 2 if (li > 1 and b == 0) else 1, dev))
                b += 1
            self.layers.append(blocks)
        self.return_layers = tuple(return_layers)
        self.num_channels = [sd[f"layer{li}.0.conv3.weight"].shape[0] for li in self.return_layers]

    @torch.no_grad()
    def __call__(self, images):
        x = self.stem(to_nhwc_bf16(images))
        x = max_pool_nhwc(x, 3, 2, 1)
        outs = []
        for li, blocks in enumerate(self.layers, start=1):
            for blk in blocks:
                x = blk(x)
            if li in self.return_layers:
                outs.append(x)
        return outs


class InputProj:
    """richsem.py:295-310 / :593-612: ``input_proj[l]`` = Conv2d(C_l, 256, 1) + GroupNorm(32, 256) on stage l, and for every further
    level Conv2d(., 256, 3, stride 2, padding 1) + GroupNorm on the last stage's map (then on the previous extra level).  Built from the
    ``input_proj.*`` entries of the model's state_dict (``input_proj.{l}.0.weight / .bias`` convolution, ``.1.weight / .bias`` norm).
    Returns per level the projected map as the (N, H_l W_l, 256) token matrix the encoder consumes, and its (H_l, W_l)."""

    def __init__(self, state_dict, device="cuda", groups=32):
        dev = torch.device(device)
        sd = {k: v.detach().to(dev) for k, v in state_dict.items()}
        self.convs, self.norms, self.groups = [], [], groups
        l = 0
        while f"{l}.0.weight" in sd:
            w = sd[f"{l}.0.weight"]
            k = w.shape[-1]
            self.convs.append(ConvAffine(w, None, sd[f"{l}.0.bias"], stride=1 if k == 1 else 2, padding=0 if k == 1 else 1, relu=False))
            self.norms.append((sd[f"{l}.1.weight"].float(), sd[f"{l}.1.bias"].float()))
            l += 1

    @torch.no_grad()
    def __call__(self, features, out_dtype=torch.float32):
        srcs, shapes = [], []
        n_stage = len(features)
        for l, conv in enumerate(self.convs):
            if l < n_stage:
                y = conv(features[l])
            elif l == n_stage:
                y = conv(features[-1])                                   # richsem.py:604
            else:
                y = conv(prev)                                           # :606 (the previous projected level, after its norm)
            N, H, W, C = y.shape
            if C == 8 * self.groups:      # RichSem's GroupNorm(32, 256): the library's kernel (one pixel's group = one 16-byte vector)
                need_prev = l + 1 < len(self.convs) and l + 1 > n_stage
                g32, prev = group_norm8_nhwc(y, self.norms[l][0], self.norms[l][1], 1e-5, want_f32=out_dtype != torch.bfloat16,
                                             want_bf16=need_prev or out_dtype == torch.bfloat16)
                tok = g32 if out_dtype != torch.bfloat16 else prev
                srcs.append(tok.view(N, H * W, C).to(out_dtype))
            else:
                g = F.group_norm(y.permute(0, 3, 1, 2).float(), self.groups, self.norms[l][0], self.norms[l][1], 1e-5)
                prev = g.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)
                srcs.append(g.permute(0, 2, 3, 1).reshape(N, H * W, C).to(out_dtype))
            shapes.append((H, W))
        return srcs, shapes


# ---- trainable form (forward + backward on the library's kernels) -----------------------------------------------------------------
from torch import nn                                         # noqa: E402

from .conv import ConvAffineFunction, GroupNorm8Function, PackCache, conv_dgrad, conv_forward, conv_wgrad_group              # noqa: E402


class FrozenBatchNorm2d(nn.Module):
    """Buffers of the reference's FrozenBatchNorm2d (models/richsem/backbone.py:20-44: weight, bias, running_mean, running_var; a
    ``num_batches_tracked`` entry in a loaded state_dict is dropped as the reference does).  The affine itself runs in the preceding
    convolution's epilogue; ``scale_shift`` is backbone.py:51-55."""

    def __init__(self, n):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        state_dict.pop(prefix + "num_batches_tracked", None)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def scale_shift(self):
        ver = (self.weight._version, self.bias._version, self.running_mean._version, self.running_var._version, self.weight.data_ptr())
        if getattr(self, "_folded_ver", None) != ver:       # the buffers are frozen: folded once (and again after a load_state_dict)
            self._folded = fold_bn(self.weight, self.bias, self.running_mean, self.running_var, 1e-5)
            self._folded_ver = ver
        return self._folded


class _ConvWeight(nn.Module):
    """the parameter of a bias-free nn.Conv2d under its own name (``<name>.weight``)"""

    def __init__(self, cin, cout, k, stride=1, padding=0):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")
        self.stride, self.padding = stride, padding
        self.pack_cache = PackCache()


def _conv_bn_act(x, conv, bn, relu, residual=None):
    scale, shift = bn.scale_shift()
    return ConvAffineFunction.apply(x, conv.weight, scale, shift, residual, conv.stride, conv.padding, relu, conv.pack_cache)


class BottleneckFunction(torch.autograd.Function):
    """One bottleneck block (torchvision's Bottleneck.forward with the frozen norms folded: conv1 -> relu -> conv2 -> relu -> conv3 (+ identity
    or downsample(x)) -> relu) as ONE autograd node, so that the element-wise work of its backward runs in the convolutions' epilogues
    (``msda_conv_dgrad_fused_bf16``): the ReLU masks of conv1 / conv2 are applied by the input gradient that produces their gradient, the
    two branches' gradients of x are summed by conv1's input gradient.  Per block that removes three ``threshold_backward`` passes and one or
    two adds over the block's largest tensors.  ``mask_input``: x is itself the output of a ReLU whose node does NOT mask (the previous
    block, told so by ``masked_by_consumer``): dx comes back as the gradient at that ReLU's input."""

    @staticmethod
    def forward(ctx, x, w1, w2, w3, wd, blk, mask_input, masked_by_consumer):
        c1, c2, c3 = blk.conv1, blk.conv2, blk.conv3
        (s1, b1), (s2, b2), (s3, b3) = blk.bn1.scale_shift(), blk.bn2.scale_shift(), blk.bn3.scale_shift()
        x = x.contiguous()
        planes = w1.shape[0]
        o1 = conv_forward(x, c1.pack_cache.get(w1, s1, False), s1, b1, None, planes, 1, 1, 1, 0, True)
        o2 = conv_forward(o1, c2.pack_cache.get(w2, s2, False), s2, b2, None, planes, 3, 3, c2.stride, 1, True)
        if wd is not None:
            cd = blk.downsample[0]
            sd_, bd_ = blk.downsample[1].scale_shift()
            idt = conv_forward(x, cd.pack_cache.get(wd, sd_, False), sd_, bd_, None, 4 * planes, 1, 1, cd.stride, 0, False)
        else:
            idt, sd_ = x, None
        y = conv_forward(o2, c3.pack_cache.get(w3, s3, False), s3, b3, idt, 4 * planes, 1, 1, 1, 0, True)
        ctx.save_for_backward(x, o1, o2, y, w1, w2, w3, wd, s1, s2, s3, sd_)
        ctx.cfg = (blk, mask_input, masked_by_consumer)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, o1, o2, y, w1, w2, w3, wd, s1, s2, s3, sd_ = ctx.saved_tensors
        blk, mask_input, masked_by_consumer = ctx.cfg
        c1, c2, c3 = blk.conv1, blk.conv2, blk.conv3
        planes = w1.shape[0]
        # gradient at the last ReLU's input (= the identity branch's gradient): masked by this node unless every consumer of y did it
        dz3 = dy.contiguous() if masked_by_consumer else torch.ops.aten.threshold_backward(dy.contiguous(), y, 0)
        need = ctx.needs_input_grad
        dz2 = conv_dgrad(dz3, c3.pack_cache.get(w3, s3, True), o2.shape, 4 * planes, 1, 1, 1, 0, relu_out=o2)
        dz1 = conv_dgrad(dz2, c2.pack_cache.get(w2, s2, True), o1.shape, planes, 3, 3, c2.stride, 1, relu_out=o1)
        dx = None
        if wd is not None:
            cd = blk.downsample[0]
            if need[0]:
                dxd = conv_dgrad(dz3, cd.pack_cache.get(wd, sd_, True), x.shape, 4 * planes, 1, 1, cd.stride, 0)
                dx = conv_dgrad(dz1, c1.pack_cache.get(w1, s1, True), x.shape, planes, 1, 1, 1, 0, add=dxd, relu_out=x if mask_input else None)
        elif need[0]:
            dx = conv_dgrad(dz1, c1.pack_cache.get(w1, s1, True), x.shape, planes, 1, 1, 1, 0, add=dz3, relu_out=x if mask_input else None)
        # the block's weight gradients: one launch of the product kernel and one of the reduction for all of them
        probs, slots = [], []
        for i, (dz, inp, cout, k, stride, pad, sc) in enumerate(((dz1, x, planes, 1, 1, 0, s1), (dz2, o1, planes, 3, c2.stride, 1, s2),
                                                                  (dz3, o2, 4 * planes, 1, 1, 0, s3),
                                                                  (dz3, x, 4 * planes, 1, blk.downsample[0].stride if wd is not None else 1, 0, sd_))):
            if need[1 + i] and (i < 3 or wd is not None):
                probs.append((dz, inp, cout, k, k, stride, pad, sc))
                slots.append(i)
        grads = [None] * 4
        if probs:
            for i, dw in zip(slots, conv_wgrad_group(probs)):
                grads[i] = dw
        dw1, dw2, dw3, dwd = grads
        return dx, dw1, dw2, dw3, dwd, None, None, None


class Bottleneck(nn.Module):
    expansion = 4
    fused = True          # forward + backward as one autograd node (BottleneckFunction) where the channel counts allow it

    def __init__(self, inplanes, planes, stride=1, downsample=False):
        super().__init__()
        self.conv1, self.bn1 = _ConvWeight(inplanes, planes, 1), FrozenBatchNorm2d(planes)
        self.conv2, self.bn2 = _ConvWeight(planes, planes, 3, stride, 1), FrozenBatchNorm2d(planes)
        self.conv3, self.bn3 = _ConvWeight(planes, planes * 4, 1), FrozenBatchNorm2d(planes * 4)
        self.downsample = nn.Sequential(_ConvWeight(inplanes, planes * 4, 1, stride), FrozenBatchNorm2d(planes * 4)) if downsample else None

    def forward(self, x, mask_input=False, masked_by_consumer=False):
        """``mask_input`` / ``masked_by_consumer``: see BottleneckFunction (set by ResNet50.forward for blocks inside a stage; both False is
        always correct)"""
        planes, inplanes = self.conv1.weight.shape[:2]
        if Bottleneck.fused and torch.is_grad_enabled() and self.conv3.weight.requires_grad and planes % 128 == 0 and inplanes % 128 == 0:
            return BottleneckFunction.apply(x, self.conv1.weight, self.conv2.weight, self.conv3.weight,
                                            self.downsample[0].weight if self.downsample is not None else None, self, mask_input,
                                            masked_by_consumer)
        identity = x if self.downsample is None else _conv_bn_act(x, self.downsample[0], self.downsample[1], False)
        out = _conv_bn_act(x, self.conv1, self.bn1, True)
        out = _conv_bn_act(out, self.conv2, self.bn2, True)
        return _conv_bn_act(out, self.conv3, self.bn3, True, residual=identity)


class ResNet50(nn.Module):
    """Trainable ResNet-50 body with torchvision's module / parameter names (so ``load_state_dict`` takes its checkpoint and the
    reference's ``'layer2' / 'layer3' / 'layer4' in name`` freezing rule applies, backbone.py:65-67), every convolution through
    ConvAffineFunction: forward, input gradient and weight gradient on the library's MFMA kernels, NHWC bf16 activations, fp32 master
    weights.  ``forward(images)`` returns the NHWC bf16 outputs of ``return_layers``."""

    def __init__(self, layers=(3, 4, 6, 3), width=64, return_layers=(2, 3, 4), train_backbone=True):
        super().__init__()
        self.conv1, self.bn1 = _ConvWeight(3, width, 7, 2, 3), FrozenBatchNorm2d(width)
        inplanes = width
        for li, (n, planes) in enumerate(zip(layers, (width, width * 2, width * 4, width * 8)), start=1):
            blocks = []
            for b in range(n):
                blocks.append(Bottleneck(inplanes, planes, 2 if (li > 1 and b == 0) else 1, downsample=b == 0))
                inplanes = planes * 4
            setattr(self, f"layer{li}", nn.Sequential(*blocks))
        self.return_layers = tuple(return_layers)
        self.num_channels = [width * 2 ** (li - 1) * 4 for li in self.return_layers]
        for name, p in self.named_parameters():          # backbone.py:65-67
            if not train_backbone or ("layer2" not in name and "layer3" not in name and "layer4" not in name):
                p.requires_grad_(False)

    def forward(self, images):
        x = _conv_bn_act(to_nhwc_bf16(images), self.conv1, self.bn1, True)
        if x.requires_grad:      # an unfrozen stem (the reference's total_finetune): PyTorch's pooling, which has a backward
            x = F.max_pool2d(x.permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1).contiguous()
        else:                    # the default: stem and layer1 frozen (backbone.py:65-67), no gradient flows through the pool
            x = max_pool_nhwc(x, 3, 2, 1)
        outs, prev_fused = [], False
        for li in range(1, 5):
            blocks = getattr(self, f"layer{li}")
            for b, blk in enumerate(blocks):
                # inside a stage a block's output feeds the next block alone: that block's node masks the gradient it returns with this
                # block's ReLU (in conv1's input-gradient epilogue), and this block's node then skips its own mask pass.  Both blocks must
                # run as BottleneckFunction nodes for the pair of flags to be set.
                fused_here = Bottleneck.fused and torch.is_grad_enabled() and blk.conv3.weight.requires_grad and blk.conv1.weight.shape[0] % 128 == 0 \
                    and blk.conv1.weight.shape[1] % 128 == 0
                nxt = blocks[b + 1] if b + 1 < len(blocks) else None
                fused_next = nxt is not None and Bottleneck.fused and torch.is_grad_enabled() and nxt.conv3.weight.requires_grad \
                    and nxt.conv1.weight.shape[0] % 128 == 0 and nxt.conv1.weight.shape[1] % 128 == 0
                x = blk(x, mask_input=b > 0 and fused_here and prev_fused, masked_by_consumer=fused_here and fused_next)
                prev_fused = fused_here
            if li in self.return_layers:
                outs.append(x)
        return outs


class _ProjConv(nn.Module):
    """the convolution of one input projection under nn.Conv2d's parameter names (``weight``, ``bias``)"""

    def __init__(self, cin, cout, k, stride, padding):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.zeros(cout))
        nn.init.xavier_uniform_(self.weight, gain=1)           # richsem.py:385-387
        self.stride, self.padding, self.pack_cache = stride, padding, PackCache()


class InputProjection(nn.ModuleList):
    """Trainable form of the input projections (richsem.py:295-310): an ``nn.ModuleList`` of ``nn.Sequential(conv, GroupNorm(32, hidden))``
    like the reference's ``input_proj``, so the parameter names are the reference's natively (``{l}.0.weight``, ``{l}.0.bias``,
    ``{l}.1.weight``, ``{l}.1.bias``) -- as the root module and nested (``model.input_proj = InputProjection()`` gives
    ``input_proj.{l}.0.weight``: a reference checkpoint loads with ``strict=True``).  The convolutions run ConvAffineFunction (bias =
    the epilogue's shift, with its gradient), the GroupNorm(32, 256) the library's NHWC kernels with their backward (conv.GroupNorm8Function;
    other group sizes: PyTorch's on the channels-first view).
    ``forward(features)`` as :class:`InputProj`."""

    def __init__(self, in_channels=(512, 1024, 2048), hidden=256, num_levels=4, groups=32):
        layers = []
        for l in range(num_levels):
            if l < len(in_channels):
                conv = _ProjConv(in_channels[l], hidden, 1, 1, 0)
            else:
                conv = _ProjConv(in_channels[-1] if l == len(in_channels) else hidden, hidden, 3, 2, 1)
            layers.append(nn.Sequential(conv, nn.GroupNorm(groups, hidden)))
        super().__init__(layers)
        self.n_stage = len(in_channels)
        self._one = None          # the epilogue's unit scale: made on first use on the input's device (not a parameter, not a buffer)

    def forward(self, features, out_dtype=torch.float32):
        srcs, shapes, prev = [], [], None
        if self._one is None or self._one.device != features[0].device:
            self._one = torch.ones(self[0][0].weight.shape[0], device=features[0].device)
        for l, (conv, norm) in enumerate(self):
            x = features[l] if l < self.n_stage else (features[-1] if l == self.n_stage else prev)
            y = ConvAffineFunction.apply(x, conv.weight, self._one, conv.bias, None, conv.stride, conv.padding, False, conv.pack_cache)
            N, H, W, C = y.shape
            if C == 8 * norm.num_groups and out_dtype == torch.bfloat16:      # the shipped GroupNorm(32, 256), bf16 consumers: the library's kernels, forward and backward, on NHWC bf16
                prev = GroupNorm8Function.apply(y, norm.weight, norm.bias, norm.eps)
                srcs.append(prev.reshape(N, H * W, C).to(out_dtype))
            else:
                g = norm(y.permute(0, 3, 1, 2).float())
                prev = g.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)
                srcs.append(g.permute(0, 2, 3, 1).reshape(N, H * W, C).to(out_dtype))
            shapes.append((H, W))
        return srcs, shapes
