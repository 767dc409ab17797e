"""The frozen CLIP ``ModifiedResNet`` teacher's forward on the MFMA convolution kernel.

Mirror of clip/model.py:94-167 (ModifiedResNet) and :10-56 (Bottleneck) as RichSem uses them: ``clip.visual(images, ret_sp=True)``
returns the stride-32 feature map (models/richsem/richsem.py:628), ``clip.visual.attnpool`` is applied to ROI features later (:753).
Built from the reference module's own ``state_dict`` (``visual.*`` keys of a CLIP checkpoint): every convolution is packed for
csrc/conv_mfma.hip with its BatchNorm (eval mode: the teacher is frozen, richsem.py:52-55) folded into the epilogue, the ReLUs and the
residual adds fused; the anti-aliasing average pools are PyTorch ops on the NHWC tensors.  bf16 storage between layers, fp32
accumulation -- a new capability (the reference runs the teacher in fp32); forward only.
"""
import torch

from .conv import ConvAffine, avg_pool_nhwc, fold_bn, to_nchw, to_nhwc_bf16
from .modules.attnpool import AttentionPool2d


def _conv_bn(sd, conv, bn, dev, stride=1, padding=0, relu=True):
    w = sd[conv + ".weight"].to(dev)
    scale, shift = fold_bn(sd[bn + ".weight"].to(dev), sd[bn + ".bias"].to(dev), sd[bn + ".running_mean"].to(dev),
                           sd[bn + ".running_var"].to(dev), 1e-5)
    return ConvAffine(w, scale, shift, stride, padding, relu)


class _Bottleneck:
    """clip/model.py:10-56: all convolutions have stride 1; an average pool follows conv2 (and precedes the downsample conv) when
    stride > 1"""

    def __init__(self, sd, prefix, stride, dev):
        self.stride = stride
        self.conv1 = _conv_bn(sd, prefix + "conv1", prefix + "bn1", dev)
        self.conv2 = _conv_bn(sd, prefix + "conv2", prefix + "bn2", dev, padding=1)
        self.conv3 = _conv_bn(sd, prefix + "conv3", prefix + "bn3", dev, relu=True)      # relu3 after the residual add: fused
        self.down = None
        if prefix + "downsample.0.weight" in sd:
            self.down = _conv_bn(sd, prefix + "downsample.0", prefix + "downsample.1", dev, relu=False)

    def __call__(self, x):
        out = self.conv2(self.conv1(x))
        out = avg_pool_nhwc(out, self.stride)
        identity = x if self.down is None else self.down(avg_pool_nhwc(x, self.stride))
        return self.conv3(out, residual=identity)          # relu(bn3(conv3(out)) + identity)


class ModifiedResNetTeacher:
    """``state_dict``: the reference ModifiedResNet's (keys conv1.weight, bn1.*, layer1.0.conv1.weight, ..., attnpool.*).
    ``heads``: attention heads of the pool (width * 32 // 64 in CLIP, clip/model.py:283)."""

    def __init__(self, state_dict, heads, device="cuda"):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("Not implemented on the CPU")
        sd = {k: v.detach() for k, v in state_dict.items()}
        self.stem = [_conv_bn(sd, "conv1", "bn1", dev, stride=2, padding=1), _conv_bn(sd, "conv2", "bn2", dev, padding=1),
                     _conv_bn(sd, "conv3", "bn3", dev, padding=1)]
        self.layers = []
        for li in range(1, 5):
            blocks, b = [], 0
            while f"layer{li}.{b}.conv1.weight" in sd:
                stride = 2 if (li > 1 and b == 0) else 1          # _make_layer(planes, blocks, stride): first block only (:126-134)
                blocks.append(_Bottleneck(sd, f"layer{li}.{b}.", stride, dev))
                b += 1
            self.layers.append(blocks)
        self.embed_dim = sd["layer4.0.conv3.weight"].shape[0]
        self.patch_size = 32
        self.attnpool = None
        if "attnpool.positional_embedding" in sd:
            n_pos, C = sd["attnpool.positional_embedding"].shape
            self.attnpool = AttentionPool2d(int(round((n_pos - 1) ** 0.5)), C, heads, sd["attnpool.c_proj.weight"].shape[0])
            self.attnpool.load_state_dict({k[len("attnpool."):]: v for k, v in sd.items() if k.startswith("attnpool.")})
            self.attnpool = self.attnpool.to(dev).eval()

    @torch.no_grad()
    def features(self, images):
        """images (N, 3, H, W) float (already CLIP-normalised) -> stride-32 feature map, NHWC bf16 (N, H/32, W/32, embed_dim)"""
        x = to_nhwc_bf16(images)
        for conv in self.stem:
            x = conv(x)
        x = avg_pool_nhwc(x, 2)                                   # clip/model.py:149
        for blocks in self.layers:
            for blk in blocks:
                x = blk(x)
        return x

    @torch.no_grad()
    def __call__(self, images, ret_sp=False):
        """the reference's ``forward(x, ret_sp)`` (clip/model.py:143-164): ``(None, feature map (N, C, H/32, W/32))`` when ``ret_sp``,
        else the attention-pooled embedding; fp32 outputs"""
        f = to_nchw(self.features(images), torch.float32)
        if ret_sp:
            return None, f
        return self.attnpool(f)
