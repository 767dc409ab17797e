"""TEST INFRASTRUCTURE (oracle) -- the detector backbone's forward with plain torch CPU ops, never imported by the product.

FrozenBatchNorm2d follows models/richsem/backbone.py:45-56.  The ResNet-50 architecture is torchvision's published v1.5 definition
(third-party, ``torchvision>=0.6.0`` unpinned in requirements.txt:5, absent from the image; call site backbone.py:144-146), the input
projections follow models/richsem/richsem.py:295-310 / :593-612.  **Parity unpinned**: neither torchvision nor the reference's model
module can be imported here and the reference holds no fixture for the backbone."""
import torch
import torch.nn.functional as F


def _fbn(x, sd, p):
    scale = sd[p + ".weight"] * (sd[p + ".running_var"] + 1e-5).rsqrt()
    bias = sd[p + ".bias"] - sd[p + ".running_mean"] * scale
    return x * scale.reshape(1, -1, 1, 1) + bias.reshape(1, -1, 1, 1)


def _bottleneck(x, sd, p, stride):
    out = torch.relu(_fbn(F.conv2d(x, sd[p + "conv1.weight"]), sd, p + "bn1"))
    out = torch.relu(_fbn(F.conv2d(out, sd[p + "conv2.weight"], stride=stride, padding=1), sd, p + "bn2"))
    out = _fbn(F.conv2d(out, sd[p + "conv3.weight"]), sd, p + "bn3")
    identity = x
    if p + "downsample.0.weight" in sd:
        identity = _fbn(F.conv2d(x, sd[p + "downsample.0.weight"], stride=stride), sd, p + "downsample.1")
    return torch.relu(out + identity)


@torch.no_grad()
def resnet_stages(x, sd, return_layers=(2, 3, 4)):
    x = torch.relu(_fbn(F.conv2d(x, sd["conv1.weight"], stride=2, padding=3), sd, "bn1"))
    x = F.max_pool2d(x, 3, 2, 1)
    outs = []
    for li in range(1, 5):
        b = 0
        while f"layer{li}.{b}.conv1.weight" in sd:
            x = _bottleneck(x, sd, f"layer{li}.{b}.", 2 if (li > 1 and b == 0) else 1)
            b += 1
        if li in return_layers:
            outs.append(x)
    return outs


@torch.no_grad()
def input_proj(features, sd, groups=32):
    """features: list of NCHW maps -> list of (N, HW, 256) token matrices + shapes (richsem.py:593-612)"""
    srcs, l = [], 0
    while f"{l}.0.weight" in sd:
        w = sd[f"{l}.0.weight"]
        k = w.shape[-1]
        src = features[l] if l < len(features) else (features[-1] if l == len(features) else prev)
        y = F.conv2d(src, w, sd[f"{l}.0.bias"], stride=1 if k == 1 else 2, padding=0 if k == 1 else 1)
        prev = F.group_norm(y, groups, sd[f"{l}.1.weight"], sd[f"{l}.1.bias"], 1e-5)
        srcs.append(prev.flatten(2).transpose(1, 2))
        l += 1
    return srcs
