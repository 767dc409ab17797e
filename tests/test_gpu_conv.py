"""GPU (-m gpu): the MFMA convolution (csrc/conv_mfma.hip) against its definition in PyTorch fp32
ops on the same bf16-rounded operands (F.conv2d on the CPU; no reference fixture exists for a single convolution -- the network-level
fixtures are in test_gpu_clip_resnet.py).  Tolerance: fp32 accumulation of exactly representable products, so the only differences
are the summation order and the final rounding to bf16: 2^-8 relative per element."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

CASES = [  # N, H, W, Cin, Cout, k, stride, pad
    (2, 13, 17, 64, 64, 1, 1, 0),
    (2, 13, 17, 64, 64, 3, 1, 1),
    (1, 20, 31, 32, 32, 3, 1, 1),        # CLIP stem conv2 (two row tiles)
    (1, 20, 31, 32, 64, 3, 1, 1),
    (2, 9, 11, 128, 512, 1, 1, 0),       # two channel blocks of 256
    (2, 15, 14, 256, 128, 3, 2, 1),      # torchvision-style strided 3 x 3
    (1, 7, 9, 512, 256, 1, 2, 0),        # strided 1 x 1 (downsample)
    (2, 21, 33, 3, 32, 3, 2, 1),         # 3-channel stem through patches
    (1, 37, 41, 3, 64, 7, 2, 3),         # torchvision stem
    (1, 8, 8, 16, 32, 3, 1, 1),          # C_in = 16 through patches
    (3, 1, 1, 64, 96, 1, 1, 0),          # C_out = 96: three blocks of 32
    (2, 5, 6, 32, 48, 3, 1, 1),          # C_out = 48: three blocks of 16
    (1, 50, 84, 1024, 256, 1, 1, 0),     # input_proj shape of C4 (richsem.py:295-303)
    (2, 25, 42, 2048, 256, 3, 2, 1),     # the extra level's stride-2 projection of C5 (richsem.py:304-310): 36 workgroups, k = 18432 -> k split
    (1, 13, 21, 512, 512, 3, 1, 1),      # layer4-like 3 x 3 on few pixels -> k split
]


@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
@pytest.mark.parametrize("epilogue", ["affine_relu", "residual_relu", "plain"])
def test_conv_against_fp32_definition(case, epilogue):
    from richsem_amd.conv import ConvAffine, to_nhwc_bf16
    N, H, W, Cin, Cout, k, stride, pad = case
    torch.manual_seed(hash(case) % 1000)
    x = torch.randn(N, Cin, H, W).to(torch.bfloat16)
    w = (torch.randn(Cout, Cin, k, k) * (Cin * k * k) ** -0.5).to(torch.bfloat16).float()
    scale = (1 + 0.3 * torch.randn(Cout)) if epilogue != "plain" else None
    shift = torch.randn(Cout) if epilogue != "plain" else None
    ref = F.conv2d(x.float(), w, stride=stride, padding=pad)
    if scale is not None:
        ref = ref * scale[None, :, None, None] + shift[None, :, None, None]
    res = None
    if epilogue == "residual_relu":
        res = torch.randn_like(ref).to(torch.bfloat16)
        ref = ref + res.float()
    if epilogue != "plain":
        ref = torch.relu(ref)
    conv = ConvAffine(w.cuda(), None if scale is None else scale.cuda(), None if shift is None else shift.cuda(), stride, pad,
                      relu=epilogue != "plain")
    got = conv(to_nhwc_bf16(x.cuda()), None if res is None else to_nhwc_bf16(res.cuda()))
    got = got.permute(0, 3, 1, 2).float().cpu()
    assert got.shape == ref.shape
    err = (got - ref).abs()
    assert float((err / (ref.abs() + 1.0)).max()) < 2 ** -7, float((err / (ref.abs() + 1.0)).max())
    assert float(err.mean()) < 4e-3 * float(ref.abs().mean() + 1e-3)
    assert "librichsem_msda.so" in open("/proc/self/maps").read()


@pytest.mark.parametrize("co_tiles,pixel_tiles", [(1, 1), (2, 3), (4, 2), (8, 1), (16, 3), (16, 1), (8, 2)])
def test_every_tile_shape_gives_the_same_result(co_tiles, pixel_tiles):
    from richsem_amd.conv import ConvAffine, set_tiling, to_nhwc_bf16
    torch.manual_seed(5)
    x = torch.randn(2, 64, 19, 23).to(torch.bfloat16)
    w = (torch.randn(256, 64, 3, 3) * 576 ** -0.5).to(torch.bfloat16).float()
    ref = torch.relu(F.conv2d(x.float(), w, stride=1, padding=1))
    conv = ConvAffine(w.cuda(), None, None, 1, 1, relu=True)
    try:
        set_tiling(co_tiles, pixel_tiles)
        got = conv(to_nhwc_bf16(x.cuda())).permute(0, 3, 1, 2).float().cpu()
    finally:
        set_tiling(0, 0)
    assert float(((got - ref).abs() / (ref.abs() + 1.0)).max()) < 2 ** -7


RING_CASES = [  # N, H, W, Cin, Cout, k, stride, pad: C_in % 64 == 0 (the ring kernel's domain)
    (2, 13, 17, 64, 64, 1, 1, 0),        # one iteration: fewer than the ring's slots
    (2, 13, 17, 128, 32, 1, 1, 0),       # two row tiles
    (2, 13, 17, 64, 64, 3, 1, 1),
    (2, 9, 11, 128, 512, 1, 1, 0),
    (2, 15, 14, 256, 128, 3, 2, 1),
    (1, 7, 9, 512, 256, 1, 2, 0),
    (1, 50, 84, 1024, 256, 1, 1, 0),
    (1, 13, 21, 512, 512, 3, 1, 1),
    (1, 5, 7, 2048, 512, 1, 1, 0),
]


@pytest.mark.parametrize("case", RING_CASES, ids=[str(c) for c in RING_CASES])
@pytest.mark.parametrize("slots", [3, 4, 6])
def test_ring_kernel_is_bit_identical_to_the_register_staged_kernel(case, slots):
    """conv_ring_kernel (both operands prefetched through LDS rings by LDS DMA) takes the products in conv_fwd_kernel's order: the same bits"""
    from richsem_amd.conv import ConvAffine, set_ring, set_tiling, to_nhwc_bf16
    N, H, W, Cin, Cout, k, stride, pad = case
    torch.manual_seed(hash(case) % 1000)
    x = to_nhwc_bf16(torch.randn(N, Cin, H, W).cuda())
    w = (torch.randn(Cout, Cin, k, k) * (Cin * k * k) ** -0.5).cuda()
    conv = ConvAffine(w, (1 + 0.3 * torch.randn(Cout)).cuda(), torch.randn(Cout).cuda(), stride, pad, relu=True)
    Ho, Wo = conv.out_hw(H, W)
    res = torch.randn(N, Ho, Wo, Cout, device="cuda").to(torch.bfloat16)
    for ct, pt in ((0, 0), (2, 1), (4, 2), (8, 1), (16, 2)):
        if ct and (Cout // 16) % ct:
            continue
        try:
            set_tiling(ct, pt)
            set_ring(-1)
            want = conv(x, res)
            set_ring(slots)
            got = conv(x, res)
        finally:
            set_ring(0)
            set_tiling(0, 0)
        if ct:
            assert torch.equal(got, want), (ct, pt, float((got.float() - want.float()).abs().max()))
        else:      # (the automatic choice may split k over workgroups that add their sums with atomics: the order of that sum is not fixed)
            assert float(((got.float() - want.float()).abs() / (want.float().abs() + 1.0)).max()) < 2 ** -7


@pytest.mark.parametrize("case", [(2, 15, 14, 256, 128, 3, 2, 1), (2, 13, 17, 64, 64, 3, 1, 1), (1, 7, 9, 512, 256, 1, 2, 0), (2, 9, 11, 128, 512, 1, 1, 0),
                                  (1, 13, 21, 512, 512, 3, 1, 1), (2, 16, 18, 128, 128, 3, 2, 1), (1, 9, 12, 64, 128, 1, 2, 0), (1, 11, 11, 64, 64, 3, 3, 1)],
                         ids=str)
@pytest.mark.parametrize("slots", [-1, 4])
def test_fused_input_gradient_epilogue(case, slots):
    """msda_conv_dgrad_fused_bf16: dx = mask(dgrad(dz) + add) against the unfused call followed by the two PyTorch ops"""
    from richsem_amd.conv import _pack_form, conv_dgrad, set_ring
    N, H, W, Cin, Cout, k, stride, pad = case
    torch.manual_seed(3)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    w = (torch.randn(Cout, Cin, k, k) * (Cout * k * k) ** -0.5).cuda()
    packed_t = _pack_form(w, torch.ones(Cout, device="cuda"), True)
    dz = torch.randn(N, Ho, Wo, Cout, device="cuda").to(torch.bfloat16)
    add = torch.randn(N, H, W, Cin, device="cuda").to(torch.bfloat16)
    act = torch.relu(torch.randn(N, H, W, Cin, device="cuda")).to(torch.bfloat16)        # a ReLU's output: about half zeros
    act.view(-1)[:7] = torch.tensor([0.0, -0.0, 1e-30, 1.0, 0.0, 3.0, -0.0], dtype=torch.bfloat16)
    try:
        set_ring(slots)
        plain = conv_dgrad(dz, packed_t, (N, H, W, Cin), Cout, k, k, stride, pad)
        fused = conv_dgrad(dz, packed_t, (N, H, W, Cin), Cout, k, k, stride, pad, add=add, relu_out=act)
        only_mask = conv_dgrad(dz, packed_t, (N, H, W, Cin), Cout, k, k, stride, pad, relu_out=act)
    finally:
        set_ring(0)
    masked = torch.ops.aten.threshold_backward(plain, act, 0)
    if slots > 0 and Cout * k * k < 4096:      # the ring kernel takes a strided gradient by parity class (only the taps that meet data): the same sums
        try:
            set_ring(-1)
            assert torch.equal(plain, conv_dgrad(dz, packed_t, (N, H, W, Cin), Cout, k, k, stride, pad))
        finally:
            set_ring(0)
    if Cin * k * k < 4096:
        assert torch.equal(only_mask, masked)
    else:      # (a long k loop on few pixels is split over workgroups that add their sums with atomics: the order of that sum is not fixed)
        assert float(((only_mask.float() - masked.float()).abs() / (masked.float().abs() + 1e-2)).max()) < 2 ** -7
        assert bool((only_mask[act <= 0] == 0).all())
    # the fused sum is rounded once (fp32 accumulator + add), the two-op form twice: one bf16 ulp apart at most
    want = torch.ops.aten.threshold_backward((plain.float() + add.float()), act.float(), 0)
    err = (fused.float() - want).abs()
    assert float((err / (plain.float().abs() + add.float().abs() + 1e-2)).max()) < 2 ** -7
    assert bool((fused[act <= 0] == 0).all())


@pytest.mark.parametrize("case", [(2, 26, 30, 128, 128, 1, 1, 0), (2, 26, 30, 256, 128, 3, 1, 1), (1, 31, 17, 128, 256, 3, 2, 1), (3, 9, 7, 512, 128, 1, 2, 0),
                                  (1, 1, 5000, 256, 384, 1, 1, 0), (1, 1, 37, 128, 128, 1, 1, 0), (1, 40, 40, 128, 128, 5, 1, 2)], ids=str)
def test_weight_gradient_ring_is_bit_identical_to_the_register_staged_kernel(case):
    """wgrad_block_ring (stages of 32 pixels three ahead through an LDS ring by LDS DMA, bias gradient as a product with ones) against
    wgrad_block: the same MFMAs over the same k-steps in the same order -- equal bits for dw; the bias sums are formed in a different order"""
    import ctypes
    from richsem_amd import _lib
    N, H, W, Cin, Cout, k, stride, pad = case
    L = _lib.load()
    torch.manual_seed(hash(case) % 997)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    x = torch.randn(N, H, W, Cin, device="cuda").to(torch.bfloat16)
    dz = torch.randn(N, Ho, Wo, Cout, device="cuda").to(torch.bfloat16)
    nb = ctypes.c_int64(0)
    _lib.check(L.msda_conv_wgrad_workspace_bytes(N, H, W, Cin, Cout, k, k, stride, pad, ctypes.byref(nb)))
    res = {}
    for ring in (1, 0):
        try:
            _lib.check(L.msda_conv_set_wgrad_ring(ring))
            dw = torch.full((Cout, k, k, Cin), float("nan"), device="cuda")
            db = torch.full((Cout,), float("nan"), device="cuda")
            ws = torch.full((max(nb.value // 4, 1),), float("nan"), device="cuda")
            _lib.check(L.msda_conv_wgrad_bf16(dz.data_ptr(), x.data_ptr(), N, H, W, Cin, Cout, k, k, stride, pad, dw.data_ptr(), db.data_ptr(), None, 0,
                                              ws.data_ptr(), _lib.raw_stream(x.device)))
            res[ring] = (dw, db)
        finally:
            _lib.check(L.msda_conv_set_wgrad_ring(1))
    assert torch.equal(res[1][0], res[0][0]), float((res[1][0] - res[0][0]).abs().max())
    want_b = dz.float().sum(dim=(0, 1, 2))
    for ring in (1, 0):
        assert float((res[ring][1] - want_b).abs().max()) <= 1e-4 * float(want_b.abs().max()) + 1e-4, ring
    # and against the definition (fp32 products of the bf16 operands, on the GPU)
    ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).float(), (Cout, Cin, k, k), dz.permute(0, 3, 1, 2).float(), stride=stride, padding=pad)
    got = res[1][0].permute(0, 3, 1, 2)
    assert float((got - ref).abs().max()) <= 2e-3 * float(ref.abs().max())


def test_grouped_weight_gradients_equal_the_single_launches():
    """msda_conv_wgrad_group_bf16 (a bottleneck block's weight gradients sharing one launch) against msda_conv_wgrad_bf16 per problem: the
    pixel chunks differ, so the sums differ in order only"""
    from richsem_amd.conv import conv_wgrad, conv_wgrad_group
    torch.manual_seed(11)
    N, H, W = 2, 26, 30
    probs = []
    for cin, cout, k, stride, pad in ((512, 128, 1, 1, 0), (128, 128, 3, 2, 1), (128, 512, 1, 1, 0), (512, 512, 1, 2, 0), (256, 128, 3, 1, 1)):
        h, w = (H, W) if (cin, k) != (128, 1) else (H // 2, W // 2)
        Ho, Wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
        x = torch.randn(N, h, w, cin, device="cuda").to(torch.bfloat16)
        dz = torch.randn(N, Ho, Wo, cout, device="cuda").to(torch.bfloat16)
        scale = (1 + 0.3 * torch.randn(cout, device="cuda")) if k == 1 else None
        probs.append((dz, x, cout, k, k, stride, pad, scale))
    for sub in (probs, probs[:1], probs[1:4]):
        got = conv_wgrad_group(sub)
        for g, (dz, x, cout, k, _, stride, pad, scale) in zip(got, sub):
            want = conv_wgrad(dz, x, cout, k, k, stride, pad, scale)
            assert g.shape == want.shape
            assert float((g - want).abs().max()) <= 2e-5 * float(want.abs().max()), float((g - want).abs().max()) / float(want.abs().max())


def test_errors():
    from richsem_amd.conv import ConvAffine
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        ConvAffine(torch.zeros(32, 32, 1, 1))
    with pytest.raises(AssertionError):
        ConvAffine(torch.zeros(24, 32, 1, 1, device="cuda"))
    with pytest.raises(RuntimeError, match="BAD_DIMS"):
        ConvAffine(torch.zeros(32, 40, 5, 5, device="cuda"))       # C_in neither a multiple of 32 nor a few-channel input
    conv = ConvAffine(torch.zeros(32, 32, 1, 1, device="cuda"))
    with pytest.raises(AssertionError):
        conv(torch.zeros(1, 4, 4, 32, device="cuda"))            # fp32 input
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        conv(torch.zeros(1, 4, 4, 32, dtype=torch.bfloat16))


GRAD_CASES = [  # N, H, W, Cin, Cout, k, stride, pad
    (2, 12, 14, 64, 64, 3, 1, 1),
    (2, 12, 14, 64, 128, 1, 1, 0),
    (1, 13, 15, 128, 64, 3, 2, 1),       # strided 3 x 3: the input gradient reads the zero-upsampled output gradient
    (2, 9, 10, 256, 512, 1, 2, 0),       # strided 1 x 1 (projection shortcut)
    (1, 8, 8, 32, 96, 3, 1, 1),          # C_in = 32 (one k-step per barrier), C_out = 96 (the input gradient's k: 3 x 32)
    (2, 14, 19, 128, 128, 3, 1, 1),      # from here: channel counts the wgrad kernel takes (multiples of 128)
    (2, 15, 17, 256, 128, 3, 2, 1),
    (3, 7, 9, 128, 256, 1, 1, 0),
    (1, 11, 13, 256, 512, 1, 2, 0),
    (2, 30, 41, 128, 128, 3, 1, 1),      # 2460 pixels: several pixel chunks per tap (split-K atomics)
    (1, 13, 21, 512, 512, 3, 1, 1),      # few pixels, long k: forward AND input gradient take the k-split form
]


def test_k_split_is_taken_where_expected():
    """the two shapes above that are meant to exercise the k-split path do (msda_conv_*_workspace_bytes > 0), a well-filled one does not"""
    import ctypes
    from richsem_amd import _lib
    L = _lib.load()
    nb = ctypes.c_int64(0)
    _lib.check(L.msda_conv_forward_workspace_bytes(2, 25, 42, 2048, 256, 3, 3, 2, 1, ctypes.byref(nb)))
    assert nb.value == 2 * 13 * 21 * 256 * 4
    _lib.check(L.msda_conv_forward_workspace_bytes(1, 13, 21, 512, 512, 3, 3, 1, 1, ctypes.byref(nb)))
    assert nb.value == 13 * 21 * 512 * 4
    _lib.check(L.msda_conv_dgrad_workspace_bytes(1, 13, 21, 512, 512, 3, 3, 1, 1, 13, 21, ctypes.byref(nb)))
    assert nb.value == 13 * 21 * 512 * 4
    _lib.check(L.msda_conv_forward_workspace_bytes(2, 100, 168, 128, 512, 1, 1, 1, 0, ctypes.byref(nb)))
    assert nb.value == 0


@pytest.mark.parametrize("case", GRAD_CASES, ids=[str(c) for c in GRAD_CASES])
@pytest.mark.parametrize("with_residual", [False, True])
def test_gradients_against_fp32_autograd(case, with_residual):
    """ConvAffineFunction (forward + input gradient on the MFMA kernel, weight gradient by the library) against torch autograd through
    the same block in fp32 on the same bf16-rounded operands.  Tolerance: bf16 rounding of dy, dz and of the results."""
    from richsem_amd.conv import ConvAffineFunction, to_nhwc_bf16
    N, H, W, Cin, Cout, k, stride, pad = case
    torch.manual_seed(3 + hash(case) % 100)
    x = torch.randn(N, Cin, H, W).to(torch.bfloat16)
    w = (torch.randn(Cout, Cin, k, k) * (Cin * k * k) ** -0.5).to(torch.bfloat16).float()
    scale, shift = 1 + 0.3 * torch.randn(Cout), 0.5 * torch.randn(Cout)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = torch.randn(N, Cout, Ho, Wo).to(torch.bfloat16) if with_residual else None
    dy = torch.randn(N, Cout, Ho, Wo).to(torch.bfloat16)
    # reference: fp32 autograd
    xr, wr = x.float().requires_grad_(True), w.clone().requires_grad_(True)
    rr = res.float().requires_grad_(True) if with_residual else None
    z = F.conv2d(xr, wr, stride=stride, padding=pad) * scale[None, :, None, None] + shift[None, :, None, None]
    if with_residual:
        z = z + rr
    torch.relu(z).backward(dy.float())
    # product
    xg = to_nhwc_bf16(x.cuda()).requires_grad_(True)
    wg = w.cuda().requires_grad_(True)
    rg = to_nhwc_bf16(res.cuda()).requires_grad_(True) if with_residual else None
    y = ConvAffineFunction.apply(xg, wg, scale.cuda(), shift.cuda(), rg, stride, pad, True)
    y.backward(to_nhwc_bf16(dy.cuda()))

    def close(got, want, name):
        err = (got - want).abs()
        s = float(want.abs().max())
        assert float(err.max()) <= 2e-2 * s and float(err.mean()) <= 3e-3 * s, (name, float(err.max()) / s, float(err.mean()) / s)

    close(xg.grad.permute(0, 3, 1, 2).float().cpu(), xr.grad, "dx")
    close(wg.grad.cpu(), wr.grad, "dw")
    if with_residual:
        close(rg.grad.permute(0, 3, 1, 2).float().cpu(), rr.grad, "dres")


@pytest.mark.parametrize("shape,k,stride,pad,is_max", [((2, 20, 30, 64), 2, 2, 0, False), ((1, 21, 33, 32), 2, 2, 0, False),
                                                      ((2, 9, 9, 8), 3, 3, 0, False), ((2, 37, 41, 64), 3, 2, 1, True),
                                                      ((1, 8, 8, 256), 3, 2, 1, True)])
def test_pooling_kernels(shape, k, stride, pad, is_max):
    """msda_pool_nhwc_bf16 against F.avg_pool2d / F.max_pool2d in fp32 on the same bf16 input (one rounding of the mean)"""
    from richsem_amd.conv import _pool
    torch.manual_seed(k)
    x = torch.randn(*shape, device="cuda").to(torch.bfloat16)
    got = _pool(x, k, stride, pad, is_max).permute(0, 3, 1, 2).float()
    xn = x.permute(0, 3, 1, 2).float()
    want = F.max_pool2d(xn, k, stride, pad) if is_max else F.avg_pool2d(xn, k, stride)
    assert got.shape == want.shape
    if is_max:
        assert torch.equal(got, want)
    else:
        assert float((got - want).abs().max()) <= 2 ** -8 * float(want.abs().max())


@pytest.mark.parametrize("shape", [(2, 12, 20, 256), (1, 100, 168, 256), (3, 2, 3, 64)])
def test_group_norm_kernel(shape):
    """msda_groupnorm8_nhwc_bf16 against F.group_norm in fp32 on the same bf16 input"""
    from richsem_amd.conv import group_norm8_nhwc
    torch.manual_seed(shape[1])
    x = (torch.randn(*shape, device="cuda") * 2 + 0.5).to(torch.bfloat16)
    C = shape[3]
    gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
    want = F.group_norm(x.permute(0, 3, 1, 2).float(), C // 8, gamma, beta, 1e-5).permute(0, 2, 3, 1)
    g32, g16 = group_norm8_nhwc(x, gamma, beta)
    assert float((g32 - want).abs().max()) <= 2e-5 * float(want.abs().max())
    assert float((g16.float() - want).abs().max()) <= 2 ** -8 * float(want.abs().max())


def test_random_shapes_forward_and_gradients():
    """24 seeded random convolutions (kernel 1 / 3 / 5, stride 1 / 2, odd sizes, every channel class the kernels take) through
    ConvAffineFunction against fp32 autograd: forward, input gradient, weight gradient (own kernel where C_in, C_out are multiples of
    128), bias-free.  Catches edge handling (padding rows, chunk tails of the pixel split, strided transposed reads)."""
    from richsem_amd.conv import ConvAffineFunction, to_nhwc_bf16
    rng = np.random.default_rng(2024)
    for trial in range(24):
        k = int(rng.choice([1, 3, 5]))
        stride = int(rng.choice([1, 2]))
        pad = int(rng.integers(0, k // 2 + 1))
        cin = int(rng.choice([32, 64, 96, 128, 256]))
        cout = int(rng.choice([32, 64, 128, 256, 384]))
        N, H, W = int(rng.integers(1, 4)), int(rng.integers(k, 40)), int(rng.integers(k, 40))
        torch.manual_seed(trial)
        x = torch.randn(N, cin, H, W).to(torch.bfloat16)
        w = (torch.randn(cout, cin, k, k) * (cin * k * k) ** -0.5).to(torch.bfloat16).float()
        scale, shift = 1 + 0.2 * torch.randn(cout), 0.3 * torch.randn(cout)
        xr, wr = x.float().requires_grad_(True), w.clone().requires_grad_(True)
        z = torch.relu(F.conv2d(xr, wr, stride=stride, padding=pad) * scale[None, :, None, None] + shift[None, :, None, None])
        dy = torch.randn_like(z).to(torch.bfloat16)
        z.backward(dy.float())
        xg, wg = to_nhwc_bf16(x.cuda()).requires_grad_(True), w.cuda().requires_grad_(True)
        y = ConvAffineFunction.apply(xg, wg, scale.cuda(), shift.cuda(), None, stride, pad, True)
        y.backward(to_nhwc_bf16(dy.cuda()))
        tag = (trial, N, H, W, cin, cout, k, stride, pad)
        for got, want, name in ((y.permute(0, 3, 1, 2).float().cpu(), z.detach(), "y"), (xg.grad.permute(0, 3, 1, 2).float().cpu(), xr.grad, "dx"),
                                (wg.grad.cpu(), wr.grad, "dw")):
            s = float(want.abs().max()) + 1e-6
            err = (got - want).abs()
            assert float(err.max()) <= 3e-2 * s and float(err.mean()) <= 4e-3 * s, (tag, name, float(err.max()) / s, float(err.mean()) / s)
