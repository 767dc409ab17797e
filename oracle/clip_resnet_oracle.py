"""TEST INFRASTRUCTURE (oracle) -- restatement of the reference's CLIP ModifiedResNet forward with plain torch CPU ops, never imported
by the product.

Follows clip/model.py:143-164 (stem: three conv-bn-relu, 2 x 2 average pool; four stages) and :42-56 (Bottleneck.forward: conv1-bn-relu,
conv2-bn-relu, avgpool(stride), conv3-bn, downsample = avgpool + conv + bn, add, relu), BatchNorm in eval mode
(y = (x - mean) / sqrt(var + 1e-5) * w + b).  Works from a state_dict; pinned by tests/golden/clip_resnet_*.npz, which
tests/golden/make_golden_clip_resnet.py generated from the reference class itself (tests/test_oracle_clip_resnet.py)."""
import torch
import torch.nn.functional as F


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)


def _bottleneck(x, sd, p, stride):
    out = torch.relu(_bn(F.conv2d(x, sd[p + "conv1.weight"]), sd, p + "bn1"))
    out = torch.relu(_bn(F.conv2d(out, sd[p + "conv2.weight"], padding=1), sd, p + "bn2"))
    if stride > 1:
        out = F.avg_pool2d(out, stride)
    out = _bn(F.conv2d(out, sd[p + "conv3.weight"]), sd, p + "bn3")
    identity = x
    if p + "downsample.0.weight" in sd:
        identity = F.avg_pool2d(x, stride) if stride > 1 else x
        identity = _bn(F.conv2d(identity, sd[p + "downsample.0.weight"]), sd, p + "downsample.1")
    return torch.relu(out + identity)


@torch.no_grad()
def feature_map(x, sd):
    """x (N, 3, H, W) -> (N, C, H/32, W/32), in x's dtype"""
    sd = {k: v.to(x.dtype) if v.is_floating_point() else v for k, v in sd.items()}
    x = torch.relu(_bn(F.conv2d(x, sd["conv1.weight"], stride=2, padding=1), sd, "bn1"))
    x = torch.relu(_bn(F.conv2d(x, sd["conv2.weight"], padding=1), sd, "bn2"))
    x = torch.relu(_bn(F.conv2d(x, sd["conv3.weight"], padding=1), sd, "bn3"))
    x = F.avg_pool2d(x, 2)
    for li in range(1, 5):
        b = 0
        while f"layer{li}.{b}.conv1.weight" in sd:
            x = _bottleneck(x, sd, f"layer{li}.{b}.", 2 if (li > 1 and b == 0) else 1)
            b += 1
    return x
