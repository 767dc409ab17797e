"""Shared by tests/golden/make_golden_clip_resnet.py (build container, with the reference class) and tests/test_gpu_clip_resnet.py:
deterministic parameters for a ModifiedResNet of reduced size, from a numpy generator (stable across platforms and versions)."""
import numpy as np
import torch

CASES = {   # name: (layers, width, heads, output_dim, input_resolution, input shape, seed)
    "clip_resnet_w64_l1111": ((1, 1, 1, 1), 64, 32, 128, 64, (2, 3, 64, 64), 11),      # CLIP-RN50's width (stem 32-32-64, 2048-wide map)
    "clip_resnet_w32_l2111": ((2, 1, 1, 1), 32, 16, 64, 96, (1, 3, 96, 160), 12),      # non-square input (ret_sp only), two blocks in layer1
}


def fill_state_dict(template, seed):
    """values for every key of `template` (a state_dict), drawn in sorted key order"""
    rng = np.random.default_rng(seed)
    out = {}
    for k in sorted(template.keys()):
        v = template[k]
        shape = tuple(v.shape)
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros(shape, dtype=v.dtype)
        elif k.endswith("running_var"):
            out[k] = torch.from_numpy(rng.uniform(0.5, 1.5, shape).astype(np.float32))
        elif k.endswith("running_mean"):
            out[k] = torch.from_numpy(rng.normal(0, 0.2, shape).astype(np.float32))
        elif ".bn" in k or k.startswith("bn") or "downsample.1" in k:
            if k.endswith("weight"):
                out[k] = torch.from_numpy(rng.uniform(0.7, 1.3, shape).astype(np.float32))
            else:
                out[k] = torch.from_numpy(rng.normal(0, 0.2, shape).astype(np.float32))
        elif len(shape) == 4:      # convolution: keeps the activations at unit scale
            fan_in = shape[1] * shape[2] * shape[3]
            out[k] = torch.from_numpy(rng.normal(0, (1.5 / fan_in) ** 0.5, shape).astype(np.float32))
        elif len(shape) == 2:      # attention pool projections / positional embedding
            out[k] = torch.from_numpy(rng.normal(0, shape[-1] ** -0.5, shape).astype(np.float32))
        else:
            out[k] = torch.from_numpy(rng.normal(0, 0.1, shape).astype(np.float32))
    return out
