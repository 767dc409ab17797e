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


def test_trainable_resnet_names_forward_and_gradients():
    """ResNet50 (nn.Module): torchvision's state_dict keys, the reference's freezing rule, forward equal to the inference form, and the
    gradients of the trained stages against fp32 autograd through the oracle's op sequence (reduced depth / size)."""
    from richsem_amd.backbone import ResNet50, ResNet50Frozen
    layers = (1, 2, 1, 1)
    sd = resnet_state_dict(layers=layers, seed=4)
    net = ResNet50(layers=layers).cuda()
    assert sorted(net.state_dict().keys()) == sorted(sd.keys())
    net.load_state_dict(sd)
    trainable = [n for n, p in net.named_parameters() if p.requires_grad]
    assert trainable and all(n.startswith(("layer2", "layer3", "layer4")) for n in trainable)
    assert all(not p.requires_grad for n, p in net.named_parameters() if n.startswith(("conv1", "layer1")))
    x = torch.from_numpy(np.random.default_rng(6).normal(0, 1, (2, 3, 64, 96)).astype(np.float32))
    outs = net(x.cuda())
    frozen = ResNet50Frozen(sd)(x.cuda())
    for a, b in zip(outs, frozen):
        assert torch.equal(a, b)
    # gradients: loss = sum_l <out_l, g_l>
    gs = [torch.from_numpy(np.random.default_rng(7 + i).normal(0, 1, tuple(o.shape)).astype(np.float32)).to(torch.bfloat16) for i, o in enumerate(outs)]
    sum((o.float() * g.cuda().float()).sum() for o, g in zip(outs, gs)).backward()
    sdr = {k: v.clone().requires_grad_(k.startswith(("layer2", "layer3", "layer4")) and v.dim() == 4) for k, v in sd.items()}
    with torch.enable_grad():
        want = BO.resnet_stages.__wrapped__(x, sdr)
        sum((w * g.float().permute(0, 3, 1, 2)).sum() for w, g in zip(want, gs)).backward()
    # yardstick: the same op sequence as PyTorch bf16 ops on the GPU (its own rounding + ReLU-mask flips against the fp32 run)
    sd16 = {k: v.cuda().to(torch.bfloat16).requires_grad_(k.startswith(("layer2", "layer3", "layer4")) and v.dim() == 4) for k, v in sd.items()}
    with torch.enable_grad():
        w16 = BO.resnet_stages.__wrapped__(x.cuda().to(torch.bfloat16), sd16)
        sum((w.float() * g.cuda().float().permute(0, 3, 1, 2)).sum() for w, g in zip(w16, gs)).backward()
    worst = 0.0
    for n, p in net.named_parameters():
        if not p.requires_grad:
            assert p.grad is None
            continue
        ref = sdr[n].grad
        s = float(ref.abs().max())
        mine = float((p.grad.cpu() - ref).abs().mean()) / s
        theirs = float((sd16[n].grad.float().cpu() - ref).abs().mean()) / s
        cos = float((p.grad.cpu() * ref).sum() / (p.grad.cpu().norm() * ref.norm() + 1e-30))
        worst = max(worst, mine)
        # bf16 activations flip a few ReLU masks against the fp32 run (single elements move by several per cent of the maximum); the
        # tensors as a whole must agree as well as PyTorch's own bf16 path does
        assert mine <= 1.5 * theirs + 1e-3 and cos >= 0.99, (n, mine, theirs, cos)
    print("worst mean weight-gradient error (of the tensor's maximum):", worst)


def test_fused_bottleneck_node_equals_the_per_convolution_nodes():
    """BottleneckFunction (ReLU masks and the branch sum in the input gradients' epilogues, the block's own mask left to the next block's node
    inside a stage) against the same network as one autograd node per convolution: equal outputs, gradients equal up to the roundings the
    fusion removes (a sum rounded once instead of twice)"""
    from richsem_amd.backbone import Bottleneck, ResNet50
    layers = (1, 3, 2, 2)
    sd = resnet_state_dict(layers=layers, seed=9)
    net = ResNet50(layers=layers).cuda()
    net.load_state_dict(sd)
    x = torch.from_numpy(np.random.default_rng(2).normal(0, 1, (2, 3, 96, 128)).astype(np.float32)).cuda()
    res = {}
    for fused in (True, False):
        Bottleneck.fused = fused
        try:
            for p in net.parameters():
                p.grad = None
            outs = net(x)
            gs = [torch.from_numpy(np.random.default_rng(7 + i).normal(0, 1, tuple(o.shape)).astype(np.float32)).to(torch.bfloat16).cuda()
                  for i, o in enumerate(outs)]
            torch.autograd.backward(outs, gs)
            res[fused] = ([o.detach().clone() for o in outs], {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None})
        finally:
            Bottleneck.fused = True
    for a, b in zip(res[True][0], res[False][0]):      # (the same forward kernels; layer4 at this size splits k over workgroups that add with
        assert float((a.float() - b.float()).abs().max()) <= 2 ** -6 * float(b.float().abs().max())      # atomics: the order of those sums is not fixed)
    assert sorted(res[True][1]) == sorted(res[False][1]) and len(res[True][1]) == sum(3 * n + 1 for n in layers[1:])
    for n, g in res[True][1].items():
        ref = res[False][1][n]
        cos = float((g * ref).sum() / (g.norm() * ref.norm() + 1e-30))
        rel = float((g - ref).abs().mean() / (ref.abs().mean() + 1e-30))
        assert cos > 0.9995 and rel < 2e-2, (n, cos, rel)


def test_trainable_input_projection_forward_and_gradients():
    """InputProjection (nn.Module, the reference's parameter names) against the inference form and fp32 autograd through the oracle's
    op sequence: convolution weight, bias and GroupNorm parameters."""
    from richsem_amd.backbone import InputProj, InputProjection
    sdp = input_proj_state_dict(channels=(128, 256, 512), seed=3)
    rng = np.random.default_rng(9)
    feats = [torch.from_numpy(rng.normal(0, 1, (2, h, w, c)).astype(np.float32)).to(torch.bfloat16).cuda()
             for (h, w), c in zip(((12, 20), (6, 10), (3, 5)), (128, 256, 512))]
    mod = InputProjection(in_channels=(128, 256, 512)).cuda()
    assert sorted(mod.state_dict().keys()) == sorted(sdp.keys())
    mod.load_state_dict(sdp)
    srcs, shapes = mod(feats)
    ref_srcs, ref_shapes = InputProj(sdp)(feats)
    assert shapes == ref_shapes
    for a, b in zip(srcs, ref_srcs):
        assert float((a - b).abs().max()) < 1e-5
    gs = [torch.from_numpy(rng.normal(0, 1, tuple(s.shape)).astype(np.float32)) for s in srcs]
    sum((s * g.cuda()).sum() for s, g in zip(srcs, gs)).backward()
    sdr = {k: v.clone().requires_grad_(True) for k, v in sdp.items()}
    with torch.enable_grad():
        want = BO.input_proj.__wrapped__([f.permute(0, 3, 1, 2).float().cpu() for f in feats], sdr)
        sum((w * g).sum() for w, g in zip(want, gs)).backward()
    for k, p in mod.state_dict(keep_vars=True).items():
        ref = sdr[k].grad
        got = dict(mod.named_parameters())[k].grad.cpu()
        s = float(ref.abs().max())
        assert float((got - ref).abs().mean()) <= 6e-3 * s and float((got - ref).abs().max()) <= 6e-2 * s, (k, float((got - ref).abs().max()) / s)


def test_packed_weights_follow_an_optimizer_step():
    """the per-parameter cache of packed weights (conv.PackCache) must notice in-place updates: forward, SGD step, forward again"""
    from richsem_amd.backbone import Bottleneck
    import torch.nn.functional as F
    torch.manual_seed(0)
    blk = Bottleneck(128, 32, stride=1, downsample=True).cuda()
    x = torch.randn(2, 9, 11, 128, device="cuda").to(torch.bfloat16)
    opt = torch.optim.SGD([p for p in blk.parameters()], lr=0.5)

    def reference(x):
        xn = x.permute(0, 3, 1, 2).float()
        def cb(t, conv, bn, relu, **kw):
            s, b = bn.scale_shift()
            y = F.conv2d(t, conv.weight.to(torch.bfloat16).float(), **kw) * s[None, :, None, None] + b[None, :, None, None]
            return torch.relu(y) if relu else y
        idt = cb(xn, blk.downsample[0], blk.downsample[1], False)
        o = cb(xn, blk.conv1, blk.bn1, True).to(torch.bfloat16).float()
        o = cb(o, blk.conv2, blk.bn2, True, padding=1).to(torch.bfloat16).float()
        return torch.relu(cb(o, blk.conv3, blk.bn3, False) + idt.to(torch.bfloat16).float())

    for step in range(2):
        y = blk(x)
        want = reference(x)
        err = float((y.permute(0, 3, 1, 2).float() - want).abs().max()) / float(want.abs().max())
        assert err < 3e-2, (step, err)
        opt.zero_grad()
        y.float().square().mean().backward()
        before = blk.conv2.weight.detach().clone()
        opt.step()
        assert not torch.equal(before, blk.conv2.weight.detach())


@pytest.mark.parametrize("N,H,W,C", [(2, 25, 42, 256), (1, 7, 9, 64), (2, 100, 168, 256)])
def test_group_norm8_function_matches_autograd(N, H, W, C):
    """GroupNorm(C // 8, C) forward + backward on the library's NHWC kernels against F.group_norm (fp32 autograd) on the same bf16 inputs"""
    import torch.nn.functional as F
    from richsem_amd.conv import GroupNorm8Function
    g = torch.Generator(device="cuda").manual_seed(5)
    x = (torch.randn(N, H, W, C, device="cuda", generator=g) * 2 + 0.5).to(torch.bfloat16).requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(C, device="cuda", generator=g)).requires_grad_(True)
    beta = (0.1 * torch.randn(C, device="cuda", generator=g)).requires_grad_(True)
    dy = torch.randn(N, H, W, C, device="cuda", generator=g).to(torch.bfloat16)
    out = GroupNorm8Function.apply(x, gamma, beta, 1e-5)
    out.backward(dy)
    xr = x.detach().float().requires_grad_(True)
    gr, br = gamma.detach().clone().requires_grad_(True), beta.detach().clone().requires_grad_(True)
    ref = F.group_norm(xr.permute(0, 3, 1, 2), C // 8, gr, br, 1e-5).permute(0, 2, 3, 1)
    ref.backward(dy.float())
    assert (out.float() - ref).abs().max() <= 2 ** -7 * ref.abs().max()                     # one bf16 rounding of the result
    assert (x.grad.float() - xr.grad).abs().max() <= 2 ** -7 * xr.grad.abs().max()
    assert (gamma.grad - gr.grad).abs().max() <= 2e-4 * gr.grad.abs().max()
    assert (beta.grad - br.grad).abs().max() <= 2e-4 * br.grad.abs().max()
