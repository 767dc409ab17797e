#!/usr/bin/env python3
"""Whole-network timing: the frozen CLIP-RN50 teacher (ModifiedResNet (3, 4, 6, 3), width 64: clip/model.py:94-167) forward to its
stride-32 feature map at the training shape (2 x 3 x 800 x 1344, richsem.py:628) on the MFMA convolution kernel, against the same
network as PyTorch ops in bf16 channels-last and fp32.  Synthetic weights (tests/clip_resnet_params.py).

    python tools/time_clip.py [--reps 5]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from clip_resnet_params import fill_state_dict                      # noqa: E402
from test_oracle_clip_resnet import template_state_dict             # noqa: E402
from richsem_amd.clip_resnet import ModifiedResNetTeacher           # noqa: E402
from richsem_amd.conv import ConvAffine                             # noqa: E402


def torch_net(sd, dtype, channels_last):
    dev = "cuda"
    P = {}
    for k, v in sd.items():
        if v.dim() == 4:
            w = v.to(dev, dtype)
            P[k] = w.contiguous(memory_format=torch.channels_last) if channels_last else w
    for k in list(sd):
        if k.endswith("running_var"):
            p = k[: -len(".running_var")]
            scale = sd[p + ".weight"] * (sd[p + ".running_var"] + 1e-5).rsqrt()
            P[p + ".s"] = scale.to(dev, dtype).reshape(1, -1, 1, 1)
            P[p + ".b"] = (sd[p + ".bias"] - sd[p + ".running_mean"] * scale).to(dev, dtype).reshape(1, -1, 1, 1)

    def cbr(x, c, b, relu=True, **kw):
        y = F.conv2d(x, P[c + ".weight"], **kw) * P[b + ".s"] + P[b + ".b"]
        return torch.relu(y) if relu else y

    def fwd(x):
        x = x.to(dtype)
        if channels_last:
            x = x.contiguous(memory_format=torch.channels_last)
        x = cbr(x, "conv1", "bn1", stride=2, padding=1)
        x = cbr(x, "conv2", "bn2", padding=1)
        x = cbr(x, "conv3", "bn3", padding=1)
        x = F.avg_pool2d(x, 2)
        for li in range(1, 5):
            b = 0
            while f"layer{li}.{b}.conv1.weight" in P:
                p, s = f"layer{li}.{b}.", 2 if (li > 1 and b == 0) else 1
                o = cbr(x, p + "conv1", p + "bn1")
                o = cbr(o, p + "conv2", p + "bn2", padding=1)
                if s > 1:
                    o = F.avg_pool2d(o, s)
                o = cbr(o, p + "conv3", p + "bn3", relu=False)
                idt = x
                if p + "downsample.0.weight" in P:
                    idt = cbr(F.avg_pool2d(x, s) if s > 1 else x, p + "downsample.0", p + "downsample.1", relu=False)
                x = torch.relu(o + idt)
                b += 1
        return x
    return fwd


def timeit(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    sd = fill_state_dict(template_state_dict((3, 4, 6, 3), 64, 32, 1024, 224), 21)
    x = torch.randn(2, 3, 800, 1344, device="cuda")
    net = ModifiedResNetTeacher(sd, heads=32)
    ref32, ref16 = torch_net(sd, torch.float32, False), torch_net(sd, torch.bfloat16, True)
    want = ref32(x)
    got = net.features(x).permute(0, 3, 1, 2).float()
    e = (got - want).abs()
    e16 = (ref16(x).float() - want).abs()
    s = float(want.abs().max())
    print(f"feature map {tuple(want.shape)}: max err {float(e.max()) / s:.3e} mean err {float(e.mean()) / s:.3e} of the map's max "
          f"(PyTorch bf16 channels-last: {float(e16.max()) / s:.3e} / {float(e16.mean()) / s:.3e})")
    ConvAffine.flop_counter = [0.0]
    net.features(x)
    flop = ConvAffine.flop_counter[0]
    ConvAffine.flop_counter = None
    t, t16, t32 = timeit(lambda: net.features(x), args.reps), timeit(lambda: ref16(x), args.reps), timeit(lambda: ref32(x), args.reps)
    print(f"CLIP-RN50 teacher forward, 2 x 800 x 1344 ({flop / 1e9:.0f} GFLOP): MFMA kernels {t:.2f} ms ({flop / t / 1e9:.0f} TFLOP/s, "
          f"{flop / t / 1e9 / 2500:.3f} of the dense bf16 peak);  PyTorch bf16 channels-last {t16:.2f} ms;  PyTorch fp32 {t32:.2f} ms")


if __name__ == "__main__":
    main()
