#!/usr/bin/env python3
"""Whole-network timing: ResNet-50 (frozen BN) forward on the MFMA convolution kernel at the training shape (2 x 3 x 800 x 1344) against
the same network as PyTorch ops (F.conv2d through MIOpen, channels-last bf16 and NCHW fp32 -- the reference's precision).

    python tools/time_backbone.py [--reps 5]
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from richsem_amd.backbone import ResNet50Frozen   # noqa: E402


from richsem_amd.workload import resnet50_state_dict as state_dict   # noqa: E402


def torch_net(sd, dtype, channels_last):
    """the same forward with library convolutions; frozen BN as scale / shift tensors (backbone.py:45-56)"""
    dev = "cuda"
    P = {}
    for k, v in sd.items():
        if k.endswith(".weight") and v.dim() == 4:
            w = v.to(dev, dtype)
            P[k] = w.contiguous(memory_format=torch.channels_last) if channels_last else w
    for k in list(sd):
        if k.endswith("running_var"):
            p = k[: -len(".running_var")]
            scale = sd[p + ".weight"] * (sd[p + ".running_var"] + 1e-5).rsqrt()
            P[p + ".s"] = scale.to(dev, dtype).reshape(1, -1, 1, 1)
            P[p + ".b"] = (sd[p + ".bias"] - sd[p + ".running_mean"] * scale).to(dev, dtype).reshape(1, -1, 1, 1)

    def fbn(x, p):
        return x * P[p + ".s"] + P[p + ".b"]

    def fwd(x):
        x = x.to(dtype)
        if channels_last:
            x = x.contiguous(memory_format=torch.channels_last)
        x = torch.relu(fbn(F.conv2d(x, P["conv1.weight"], stride=2, padding=3), "bn1"))
        x = F.max_pool2d(x, 3, 2, 1)
        outs = []
        for li in range(1, 5):
            b = 0
            while f"layer{li}.{b}.conv1.weight" in P:
                p = f"layer{li}.{b}."
                s = 2 if (li > 1 and b == 0) else 1
                o = torch.relu(fbn(F.conv2d(x, P[p + "conv1.weight"]), p + "bn1"))
                o = torch.relu(fbn(F.conv2d(o, P[p + "conv2.weight"], stride=s, padding=1), p + "bn2"))
                o = fbn(F.conv2d(o, P[p + "conv3.weight"]), p + "bn3")
                idt = fbn(F.conv2d(x, P[p + "downsample.0.weight"], stride=s), p + "downsample.1") if b == 0 else x
                x = torch.relu(o + idt)
                b += 1
            if li > 1:
                outs.append(x)
        return outs
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


def train(args):
    """forward + backward (layer2-4 trainable, stem and layer1 frozen: backbone.py:65-67) against PyTorch bf16 channels-last autograd"""
    from richsem_amd.backbone import ResNet50
    from richsem_amd.conv import ConvAffineFunction
    sd = state_dict()
    x = torch.randn(2, 3, 800, 1344, device="cuda")
    net = ResNet50().cuda()
    net.load_state_dict(sd)
    outs = net(x)
    gs = [torch.randn_like(o) for o in outs]

    def step():
        for p in net.parameters():
            p.grad = None
        torch.autograd.backward(net(x), gs)

    # PyTorch: same graph with library ops, bf16 channels-last, same trainable set
    P = {}
    for k, v in sd.items():
        if v.dim() == 4:
            P[k] = v.cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(k.startswith(("layer2", "layer3", "layer4")))
    for k in list(sd):
        if k.endswith("running_var"):
            p = k[: -len(".running_var")]
            scale = sd[p + ".weight"] * (sd[p + ".running_var"] + 1e-5).rsqrt()
            P[p + ".s"] = scale.cuda().to(torch.bfloat16).reshape(1, -1, 1, 1)
            P[p + ".b"] = (sd[p + ".bias"] - sd[p + ".running_mean"] * scale).cuda().to(torch.bfloat16).reshape(1, -1, 1, 1)
    gs_t = [g.permute(0, 3, 1, 2) for g in gs]

    def fwd_t(x):
        y = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        y = torch.relu(F.conv2d(y, P["conv1.weight"], stride=2, padding=3) * P["bn1.s"] + P["bn1.b"])
        y = F.max_pool2d(y, 3, 2, 1)
        outs = []
        for li in range(1, 5):
            b = 0
            while f"layer{li}.{b}.conv1.weight" in P:
                p, s = f"layer{li}.{b}.", 2 if (li > 1 and b == 0) else 1
                o = torch.relu(F.conv2d(y, P[p + "conv1.weight"]) * P[p + "bn1.s"] + P[p + "bn1.b"])
                o = torch.relu(F.conv2d(o, P[p + "conv2.weight"], stride=s, padding=1) * P[p + "bn2.s"] + P[p + "bn2.b"])
                o = F.conv2d(o, P[p + "conv3.weight"]) * P[p + "bn3.s"] + P[p + "bn3.b"]
                idt = (F.conv2d(y, P[p + "downsample.0.weight"], stride=s) * P[p + "downsample.1.s"] + P[p + "downsample.1.b"]) if b == 0 else y
                y = torch.relu(o + idt)
                b += 1
            if li > 1:
                outs.append(y)
        return outs

    def step_t():
        for v in P.values():
            v.grad = None
        torch.autograd.backward(fwd_t(x), gs_t)

    t = timeit(step, args.reps)
    if args.ours_only:
        print(f"forward + backward {t:.2f} ms")
        return
    from richsem_amd.backbone import Bottleneck
    Bottleneck.fused = False                      # one autograd node per convolution (the form before round 4) ...
    t_nodes = timeit(step, args.reps)
    ConvAffineFunction.library_wgrad = True       # ... and that form with MIOpen's weight gradients
    t_lib = timeit(step, args.reps)
    ConvAffineFunction.library_wgrad = False
    Bottleneck.fused = True
    t_t = timeit(step_t, args.reps)
    t_f = timeit(lambda: net(x), args.reps)
    print(f"ResNet-50 at 2 x 800 x 1344, layer2-4 trained: forward + backward on the library's kernels {t:.2f} ms (forward alone {t_f:.2f}; "
          f"one autograd node per convolution {t_nodes:.2f}, that with MIOpen weight gradients {t_lib:.2f});  PyTorch bf16 channels-last autograd {t_t:.2f} ms")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--ours-only", action="store_true", help="with --train: skip the PyTorch side (for rocprofv3 runs)")
    ap.add_argument("--train", action="store_true", help="forward + backward of the trained stages (layer2-4) instead of the forward")
    args = ap.parse_args()
    if args.train:
        return train(args)
    sd = state_dict()
    x = torch.randn(2, 3, 800, 1344, device="cuda")
    net = ResNet50Frozen(sd)
    ref32 = torch_net(sd, torch.float32, False)
    ref16 = torch_net(sd, torch.bfloat16, True)
    want = ref32(x)
    got = net(x)
    for g, w, n in zip(got, want, ("C3", "C4", "C5")):
        e = (g.permute(0, 3, 1, 2).float() - w).abs()
        print(f"{n}: {tuple(w.shape)}  max err {float(e.max()) / float(w.abs().max()):.3e}  mean err {float(e.mean()) / float(w.abs().max()):.3e} (of the map's max)")
    e16 = [(a.float() - w).abs().mean().item() / w.abs().max().item() for a, w in zip(ref16(x), want)]
    print("PyTorch bf16 channels-last against fp32: mean err", " ".join(f"{v:.3e}" for v in e16))
    from richsem_amd.conv import ConvAffine
    ConvAffine.flop_counter = [0.0]
    net(x)
    flop = ConvAffine.flop_counter[0]
    ConvAffine.flop_counter = None
    t = timeit(lambda: net(x), args.reps)
    t16 = timeit(lambda: ref16(x), args.reps)
    t32 = timeit(lambda: ref32(x), args.reps)
    print(f"ResNet-50 forward, 2 x 800 x 1344 (~{flop / 1e9:.0f} GFLOP): MFMA kernels {t:.2f} ms ({flop / t / 1e9:.0f} TFLOP/s, "
          f"{flop / t / 1e9 / 2500:.3f} of the dense bf16 peak);  PyTorch bf16 channels-last {t16:.2f} ms;  PyTorch fp32 {t32:.2f} ms")


if __name__ == "__main__":
    main()
