"""GPU (-m gpu): ResNet-50 with frozen BatchNorm and the input projections on the MFMA convolution kernel (richsem_amd/backbone.py,
SURVEY.md section 8a row a10) against the torch-CPU oracle (oracle/backbone_oracle.py; parity unpinned, see its header) on seeded
random weights.  Tolerance: bf16 storage through 53 convolutions, fp32 accumulation."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import backbone_oracle as BO          # noqa: E402

pytestmark = pytest.mark.gpu


def resnet_state_dict(layers=(3, 4, 6, 3), width=64, seed=0):
    """torchvision resnet bottleneck state_dict shapes, seeded values (convolutions at He scale, BN statistics spread)"""
    rng = np.random.default_rng(seed)
    sd = {}

    def conv(name, co, ci, k):
        sd[name + ".weight"] = torch.from_numpy(rng.normal(0, (1.2 / (ci * k * k)) ** 0.5, (co, ci, k, k)).astype(np.float32))

    def bn(name, c, last=False):
        sd[name + ".weight"] = torch.from_numpy(rng.uniform(0.2 if last else 0.7, 0.5 if last else 1.3, c).astype(np.float32))
        sd[name + ".bias"] = torch.from_numpy(rng.normal(0, 0.1, c).astype(np.float32))
        sd[name + ".running_mean"] = torch.from_numpy(rng.normal(0, 0.1, c).astype(np.float32))
        sd[name + ".running_var"] = torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32))

    conv("conv1", width, 3, 7); bn("bn1", width)
    inplanes = width
    for li, (n, planes) in enumerate(zip(layers, (width, width * 2, width * 4, width * 8)), start=1):
        for b in range(n):
            p = f"layer{li}.{b}."
            conv(p + "conv1", planes, inplanes, 1); bn(p + "bn1", planes)
            conv(p + "conv2", planes, planes, 3); bn(p + "bn2", planes)
            conv(p + "conv3", planes * 4, planes, 1); bn(p + "bn3", planes * 4, last=True)
            if b == 0:
                conv(p + "downsample.0", planes * 4, inplanes, 1); bn(p + "downsample.1", planes * 4)
            inplanes = planes * 4
    return sd


def input_proj_state_dict(channels=(512, 1024, 2048), hidden=256, levels=4, seed=1):
    rng = np.random.default_rng(seed)
    sd = {}
    for l in range(levels):
        ci, k = (channels[l], 1) if l < len(channels) else ((channels[-1] if l == len(channels) else hidden), 3)
        sd[f"{l}.0.weight"] = torch.from_numpy(rng.normal(0, (1.0 / (ci * k * k)) ** 0.5, (hidden, ci, k, k)).astype(np.float32))
        sd[f"{l}.0.bias"] = torch.from_numpy(rng.normal(0, 0.1, hidden).astype(np.float32))
        sd[f"{l}.1.weight"] = torch.from_numpy(rng.uniform(0.7, 1.3, hidden).astype(np.float32))
        sd[f"{l}.1.bias"] = torch.from_numpy(rng.normal(0, 0.1, hidden).astype(np.float32))
    return sd


def check(got, want, max_tol, mean_tol):
    scale = float(want.abs().max())
    err = (got - want).abs()
    assert float(err.max()) <= max_tol * scale, (float(err.max()) / scale, float(err.mean()) / scale)
    assert float(err.mean()) <= mean_tol * scale, (float(err.max()) / scale, float(err.mean()) / scale)


def test_resnet50_stages_and_input_proj_against_oracle():
    from richsem_amd.backbone import InputProj, ResNet50Frozen
    sd, sdp = resnet_state_dict(), input_proj_state_dict()
    x = torch.from_numpy(np.random.default_rng(2).normal(0, 1, (2, 3, 96, 160)).astype(np.float32))
    want = BO.resnet_stages(x, sd)
    net = ResNet50Frozen(sd)
    got = net(x.cuda())
    assert net.num_channels == [512, 1024, 2048] and len(got) == 3
    for g, w in zip(got, want):
        assert g.dtype == torch.bfloat16 and g.shape == (w.shape[0], w.shape[2], w.shape[3], w.shape[1])      # NHWC
        check(g.permute(0, 3, 1, 2).float().cpu(), w, 6e-2, 6e-3)
    # input projections on the product's own stage outputs against the oracle's projections of the same (bf16) maps
    proj = InputProj(sdp)
    srcs, shapes = proj(got)
    feats = [g.permute(0, 3, 1, 2).float().cpu() for g in got]
    want_srcs = BO.input_proj(feats, sdp)
    assert shapes == [(12, 20), (6, 10), (3, 5), (2, 3)] and len(srcs) == 4
    for s, w in zip(srcs, want_srcs):
        assert s.shape == w.shape and s.dtype == torch.float32
        check(s.cpu(), w, 3e-2, 4e-3)
    assert "librichsem_msda.so" in open("/proc/self/maps").read()
