"""Shared by tests/golden/make_golden_layers.py (build container, with the reference's classes) and tests/test_gpu_layers.py: deterministic
parameters for the d_model = 256 / 8-head layer fixtures, a pure function of (seed, state_dict key, shape) from a numpy generator (stable
across platforms and torch versions) -- so that the fixtures store inputs, outputs and gradients but not 0.4-1.7 M parameters each."""
import zlib

import numpy as np
import torch

D_MODEL, HEADS, D_FFN, LEVELS, POINTS = 256, 8, 512, 4, 4
SHAPES = [(8, 12), (4, 6), (2, 3), (1, 2)]          # S = 128


def fill(module, seed, dtype=torch.float64):
    """every parameter of `module`, in place (its state_dict keys are the reference's: the mirrors keep them)"""
    with torch.no_grad():
        for name, p in module.named_parameters():
            rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
            shape = tuple(p.shape)
            if p.dim() >= 2:
                v = rng.normal(0.0, shape[-1] ** -0.5, shape)
                if name.endswith("sampling_offsets.weight"):
                    v *= 0.3                                         # offsets of about a pixel around the bias below
            elif "norm" in name and name.endswith("weight"):
                v = 1.0 + 0.1 * rng.normal(0.0, 1.0, shape)
            elif name.endswith("sampling_offsets.bias"):
                v = rng.normal(0.0, 1.0, shape)                      # (pixels of the sampled level)
            else:
                v = 0.1 * rng.normal(0.0, 1.0, shape)
            p.copy_(torch.from_numpy(v).to(dtype))
    return module


def pack_grads(module, keep=None):
    """parameter gradients as float16 mantissas with a per-tensor scale (the fixtures feed bf16-tolerance comparisons: 11 bits suffice)"""
    out = {}
    for name, p in module.named_parameters():
        if p.grad is None or (keep is not None and not keep(name)):
            continue
        g = p.grad.detach().double().numpy()
        s = float(np.abs(g).max()) or 1.0
        out["pgrad." + name] = (g / s).astype(np.float16)
        out["pscale." + name] = np.float64(s)
    return out


def unpack_grads(z):
    return {k[len("pgrad."):]: torch.from_numpy(z[k].astype(np.float64) * float(z["pscale." + k[len("pgrad."):]])) for k in z.files
            if k.startswith("pgrad.")}
