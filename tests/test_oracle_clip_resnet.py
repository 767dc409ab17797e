"""CPU: the CLIP-ResNet oracle against the fixtures generated from the reference's own class (clip/model.py:94-167 through
tests/golden/make_golden_clip_resnet.py), with the attention-pool oracle for the pooled embedding."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import attnpool_oracle as AO, clip_resnet_oracle as RO      # noqa: E402
from clip_resnet_params import CASES, fill_state_dict                   # noqa: E402


def template_state_dict(layers, width, heads, out_dim, res):
    """shapes of the reference ModifiedResNet's state_dict (clip/model.py:103-127, :13-40), without the reference"""
    sd = {}

    def conv(name, co, ci, k):
        sd[name + ".weight"] = torch.empty(co, ci, k, k)

    def bn(name, c):
        for s in ("weight", "bias", "running_mean", "running_var"):
            sd[f"{name}.{s}"] = torch.empty(c)
        sd[name + ".num_batches_tracked"] = torch.empty((), dtype=torch.int64)

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
    sd["attnpool.positional_embedding"] = torch.empty((res // 32) ** 2 + 1, C)
    for n, o in (("k", C), ("q", C), ("v", C), ("c", out_dim)):
        sd[f"attnpool.{n}_proj.weight"] = torch.empty(o, C)
        sd[f"attnpool.{n}_proj.bias"] = torch.empty(o)
    return sd


def case_state_dict(name):
    layers, width, heads, out_dim, res, shape, seed = CASES[name]
    return fill_state_dict(template_state_dict(layers, width, heads, out_dim, res), seed)


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_reference_outputs(name):
    z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    sd = case_state_dict(name)
    fmap = RO.feature_map(torch.from_numpy(z["x"]), sd).numpy()
    assert fmap.shape == z["fmap"].shape
    assert np.abs(fmap - z["fmap"]).max() <= 2e-5 * np.abs(z["fmap"]).max()
    if "embed" in z.files:
        params = {k[len("attnpool."):]: v.numpy() for k, v in sd.items() if k.startswith("attnpool.")}
        emb = AO.attnpool(z["fmap"], params, CASES[name][2])
        assert np.abs(emb - z["embed"]).max() <= 2e-5 * np.abs(z["embed"]).max()
