"""Synthetic MSDeformAttn workloads at the shapes of BASELINE.json's configs, and the algorithmic
byte counts the roofline is quoted on.

Shapes (R50 4-scale, 1333x800 padded to 800x1344; reference util/misc.py:410-414, richsem.py:304-308):
levels 100x168, 50x84, 25x42, 13x21 -> S = 22323; N = 2 images per GPU, M = 8 heads, D = 32, L = P = 4.
  E   encoder self-attention call   Lq = S = 22323          (deformable_transformer.py:870)
  Dd  decoder cross-attention call  Lq = 1092 = 900 + 192   (deformable_transformer.py:1017)
  Em  ImageNet-LVIS mosaic step     1280x1280 -> 160^2, 80^2, 40^2, 20^2, S = Lq = 34000
One training step runs 6 E and 6 Dd calls forward and backward (6 encoder + 6 decoder layers,
config/RichSem/baseline_4scale.py:36-37).
"""
import math
from dataclasses import dataclass

import torch


def pyramid_shapes(height, width, n_levels=4):
    """Feature-map sizes of the R50 4-scale pyramid for a padded (height, width) image: strides 8/16/32
    (ceil), then each extra level is a 3x3 stride-2 pad-1 conv of the previous one (richsem.py:304-308)."""
    shapes = [(math.ceil(height / s), math.ceil(width / s)) for s in (8, 16, 32)][:n_levels]
    while len(shapes) < n_levels:
        h, w = shapes[-1]
        shapes.append(((h - 1) // 2 + 1, (w - 1) // 2 + 1))
    return shapes


@dataclass
class Call:
    name: str
    N: int
    M: int
    D: int
    P: int
    shapes: list          # [(H, W)] per level
    Lq: int
    encoder: bool         # queries are the pixels of the pyramid (Lq == S)

    @property
    def L(self):
        return len(self.shapes)

    @property
    def S(self):
        return sum(h * w for h, w in self.shapes)

    def bytes_fwd(self, e_v=4, e_s=4):
        """Algorithmic bytes of one forward call: value read once, out written once, loc + attn read."""
        return e_v * (self.N * self.S * self.M * self.D + self.N * self.Lq * self.M * self.D) \
            + e_s * 3 * self.N * self.Lq * self.M * self.L * self.P

    def bytes_bwd(self, e_v=4, e_s=4):
        """Algorithmic bytes of one backward call: value read + grad_value written, grad_out read,
        loc + attn read, grad_loc + grad_attn written."""
        return e_v * (2 * self.N * self.S * self.M * self.D + self.N * self.Lq * self.M * self.D) \
            + e_s * 6 * self.N * self.Lq * self.M * self.L * self.P


def call_E(N=2):
    sh = pyramid_shapes(800, 1344)
    return Call("E", N, 8, 32, 4, sh, sum(h * w for h, w in sh), True)


def call_Dd(N=2, Lq=1092):
    return Call("Dd", N, 8, 32, 4, pyramid_shapes(800, 1344), Lq, False)


def call_Em(N=2):
    sh = pyramid_shapes(1280, 1280)
    return Call("Em", N, 8, 32, 4, sh, sum(h * w for h, w in sh), True)


def shrunk(call, factor):
    """The same call on a pyramid `factor` times smaller per side (parity-test sizes)."""
    sh = [(max(1, round(h / factor)), max(1, round(w / factor))) for h, w in call.shapes]
    S = sum(h * w for h, w in sh)
    return Call(call.name + f"/{factor}", call.N, call.M, call.D, call.P, sh, S if call.encoder else
                max(8, call.Lq // (factor * factor)), call.encoder)


def level_tensors(call, device="cpu"):
    shapes = torch.as_tensor(call.shapes, dtype=torch.long, device=device)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    return shapes, lsi


def encoder_reference_points(call, dtype=torch.float32):
    """Pixel centres of every level, normalised (x, y); valid_ratio = 1 (deformable_transformer.py:512-525)."""
    refs = []
    for H, W in call.shapes:
        ys = (torch.arange(H, dtype=dtype) + 0.5) / H
        xs = (torch.arange(W, dtype=dtype) + 0.5) / W
        gy, gx = torch.meshgrid(ys, xs, indexing="ij")
        refs.append(torch.stack((gx.reshape(-1), gy.reshape(-1)), -1))
    return torch.cat(refs, 0)


def make_inputs(call, loc_mode="init", seed=0, dtype=torch.float32, device="cpu", jitter_px=1.0):
    """Seeded inputs for one call: value ~ N(0,1); attn = softmax(N(0,1)) over the L*P points;
    grad_out ~ N(0,1); sampling locations in one of two distributions:
      "init"    reference points (encoder: pixel centres; decoder: U(0.1,0.9)^2 box centres) + the module's
                initial offsets (head h: direction (cos,sin)(2 pi h/M)/max|.|, times k = 1..P pixels of the
                sampled level; ops/modules/ms_deform_attn.py:62-76) + N(0, jitter_px) pixels -- local pattern
      "uniform" U[0,1)^2 as in ops/test.py:34 -- worst-case locality
    Generated on the CPU generator (reproducible everywhere), then moved to `device`.
    Returns dict(value, shapes, lsi, loc, aw, grad_out)."""
    if loc_mode == "sigma4":   # the init pattern with 4 px of jitter (bench's middle distribution)
        loc_mode, jitter_px = "init", 4.0
    g = torch.Generator().manual_seed(seed)
    N, M, D, L, P, S, Lq = call.N, call.M, call.D, call.L, call.P, call.S, call.Lq
    value = torch.randn(N, S, M, D, generator=g, dtype=dtype)
    aw = torch.softmax(torch.randn(N, Lq, M, L * P, generator=g, dtype=dtype), -1).view(N, Lq, M, L, P)
    grad_out = torch.randn(N, Lq, M * D, generator=g, dtype=dtype)
    if loc_mode == "uniform":
        loc = torch.rand(N, Lq, M, L, P, 2, generator=g, dtype=dtype)
    elif loc_mode == "init":
        if call.encoder:
            ref = encoder_reference_points(call, dtype)[None].expand(N, Lq, 2)
        else:
            ref = torch.rand(N, Lq, 2, generator=g, dtype=dtype) * 0.8 + 0.1
        th = torch.arange(M, dtype=dtype) * (2.0 * math.pi / M)
        d = torch.stack([th.cos(), th.sin()], -1)
        d = d / d.abs().max(-1, keepdim=True)[0]
        k = torch.arange(1, P + 1, dtype=dtype)
        off = (d[:, None, None, :] * k[None, None, :, None]).expand(M, L, P, 2)
        off = off[None, None] + jitter_px * torch.randn(N, Lq, M, L, P, 2, generator=g, dtype=dtype)
        wh = torch.tensor([[w, h] for h, w in call.shapes], dtype=dtype)
        loc = ref[:, :, None, None, None, :] + off / wh[None, None, None, :, None, :]
    else:
        raise ValueError(loc_mode)
    shapes, lsi = level_tensors(call)
    t = dict(value=value, shapes=shapes, lsi=lsi, loc=loc.contiguous(), aw=aw.contiguous(), grad_out=grad_out)
    return {k: v.to(device) for k, v in t.items()}


def make_loc(call, loc_mode, seed=0, dtype=torch.float32, device="cpu"):
    """Only the sampling locations of make_inputs, drawn from their own generator, in one of the bench's three
    distributions: "init" (1 px jitter), "sigma4" (the init pattern + N(0, 4 px): 5-6 % of the points leave a
    6-px window margin -- offsets of a trained network are not 1 px) and "uniform"."""
    mode, jitter = {"init": ("init", 1.0), "sigma4": ("init", 4.0), "uniform": ("uniform", 0.0)}[loc_mode]
    g = torch.Generator().manual_seed(977 + seed)
    N, M, L, P, Lq = call.N, call.M, call.L, call.P, call.Lq
    if mode == "uniform":
        return torch.rand(N, Lq, M, L, P, 2, generator=g, dtype=dtype).to(device)
    if call.encoder:
        ref = encoder_reference_points(call, dtype)[None].expand(N, Lq, 2)
    else:
        ref = torch.rand(N, Lq, 2, generator=g, dtype=dtype) * 0.8 + 0.1
    th = torch.arange(M, dtype=dtype) * (2.0 * math.pi / M)
    d = torch.stack([th.cos(), th.sin()], -1)
    d = d / d.abs().max(-1, keepdim=True)[0]
    k = torch.arange(1, P + 1, dtype=dtype)
    off = (d[:, None, None, :] * k[None, None, :, None]).expand(M, L, P, 2)
    off = off[None, None] + jitter * torch.randn(N, Lq, M, L, P, 2, generator=g, dtype=dtype)
    wh = torch.tensor([[w, h] for h, w in call.shapes], dtype=dtype)
    return (ref[:, :, None, None, None, :] + off / wh[None, None, None, :, None, :]).contiguous().to(device)


LOC_MODES = ("init", "sigma4", "uniform")

# one training step of the hot path: (call, repetitions) -- 6 encoder layers + 6 decoder layers, each with its OWN
# tensors (deformable_transformer.py:870, :1017: every layer projects its own value and predicts its own offsets)
def training_step_calls(N=2):
    return [(call_E(N), 6), (call_Dd(N), 6)]

# trainable parameters whose gradients DDP all-reduces per step (R50 C3-C5 23.3 M, input_proj
# 5.6 M, 6 encoder layers 7.7 M, 6 decoder layers 9.3 M, heads / embeddings 1.7 M; reference main.py:204-206)
GRAD_ALLREDUCE_ELEMS = 47_600_000
DDP_BUCKET_BYTES = 25 * 1024 * 1024   # torch DDP's default bucket_cap_mb


def resnet50_state_dict(seed=0, width=64, layers=(3, 4, 6, 3)):
    """Synthetic parameters with the names and shapes of torchvision's ResNet-50 ``state_dict`` (no checkpoint is available offline):
    convolutions at He scale, frozen-BatchNorm statistics spread around (0, 1).  Used by bench.py's backbone row and tools/."""
    import numpy as np
    import torch
    rng = np.random.default_rng(seed)
    sd = {}

    def conv(name, co, ci, k):
        sd[name + ".weight"] = torch.from_numpy(rng.normal(0, (1.2 / (ci * k * k)) ** 0.5, (co, ci, k, k)).astype(np.float32))

    def bn(name, c):
        sd[name + ".weight"] = torch.from_numpy(rng.uniform(0.3, 0.9, c).astype(np.float32))
        sd[name + ".bias"] = torch.from_numpy(rng.normal(0, 0.1, c).astype(np.float32))
        sd[name + ".running_mean"] = torch.from_numpy(rng.normal(0, 0.1, c).astype(np.float32))
        sd[name + ".running_var"] = torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32))

    conv("conv1", width, 3, 7)
    bn("bn1", width)
    inplanes = width
    for li, (n, planes) in enumerate(zip(layers, (width, width * 2, width * 4, width * 8)), start=1):
        for b in range(n):
            p = f"layer{li}.{b}."
            conv(p + "conv1", planes, inplanes, 1)
            bn(p + "bn1", planes)
            conv(p + "conv2", planes, planes, 3)
            bn(p + "bn2", planes)
            conv(p + "conv3", planes * 4, planes, 1)
            bn(p + "bn3", planes * 4)
            if b == 0:
                conv(p + "downsample.0", planes * 4, inplanes, 1)
                bn(p + "downsample.1", planes * 4)
            inplanes = planes * 4
    return sd


def clip_rn50_state_dict(seed=0, layers=(3, 4, 6, 3), width=64, heads=32, out_dim=1024, res=224):
    """A CLIP ``ModifiedResNet`` state_dict (clip/model.py:94-167: three-convolution stem, anti-aliased bottlenecks, attention pool) with
    seeded synthetic values that keep the activations at unit scale -- the frozen teacher of the composed bench step (no checkpoint can
    be fetched here)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv(name, co, ci, k):
        sd[name + ".weight"] = torch.randn(co, ci, k, k, generator=g) * (1.5 / (ci * k * k)) ** 0.5

    def bn(name, c):
        sd[name + ".weight"] = torch.rand(c, generator=g) * 0.6 + 0.7
        sd[name + ".bias"] = torch.randn(c, generator=g) * 0.2
        sd[name + ".running_mean"] = torch.randn(c, generator=g) * 0.2
        sd[name + ".running_var"] = torch.rand(c, generator=g) + 0.5

    conv("conv1", width // 2, 3, 3); bn("bn1", width // 2)
    conv("conv2", width // 2, width // 2, 3); bn("bn2", width // 2)
    conv("conv3", width, width // 2, 3); bn("bn3", width)
    inplanes = width
    for li, (n, planes) in enumerate(zip(layers, (width, width * 2, width * 4, width * 8)), start=1):
        for b in range(n):
            p = f"layer{li}.{b}."
            stride = 2 if (li > 1 and b == 0) else 1
            conv(p + "conv1", planes, inplanes, 1); bn(p + "bn1", planes)
            conv(p + "conv2", planes, planes, 3); bn(p + "bn2", planes)
            conv(p + "conv3", planes * 4, planes, 1); bn(p + "bn3", planes * 4)
            if stride > 1 or inplanes != planes * 4:
                conv(p + "downsample.0", planes * 4, inplanes, 1); bn(p + "downsample.1", planes * 4)
            inplanes = planes * 4
    C = width * 32
    sd["attnpool.positional_embedding"] = torch.randn((res // 32) ** 2 + 1, C, generator=g) * C ** -0.5
    for n, o in (("k", C), ("q", C), ("v", C), ("c", out_dim)):
        sd[f"attnpool.{n}_proj.weight"] = torch.randn(o, C, generator=g) * C ** -0.5
        sd[f"attnpool.{n}_proj.bias"] = torch.randn(o, generator=g) * 0.1
    return sd
