"""GPU (-m gpu): the fused feed-forward kernel (reference forward_ffn,
models/richsem/deformable_transformer.py:862-866) against its definition in PyTorch fp32 ops.

A floating-point MFMA kernel, so the reference is a plain PyTorch fp32 computation of the same block on the same
(bf16-representable) inputs.  Tolerance: the kernel keeps everything in fp32 except the hidden activation, which it rounds to
bf16 once (like a bf16 nn.Linear) and the output (one rounding): |err| <= 2^-7 * max(1, |ref|) element-wise and 3e-3 on
average for LayerNorm-scaled outputs -- a transposed / permuted operand anywhere gives errors of order 1.
"""
import pytest
import torch
import torch.nn.functional as F

from richsem_amd import _lib
from richsem_amd.functions import FusedFFNFunction, ffn_forward_bf16, pack_w2_bf16
from ffn_module import FFN

pytestmark = pytest.mark.gpu
D = 256


def make(T, Fh, seed=0, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    x = (scale * r(T, D)).to(torch.bfloat16)
    w1 = (r(Fh, D) * D ** -0.5).to(torch.bfloat16)
    w2 = (r(D, Fh) * Fh ** -0.5).to(torch.bfloat16)
    return x, w1, 0.1 * r(Fh), w2, 0.1 * r(D), 1 + 0.1 * r(D), 0.1 * r(D)


def ref_fp32(x, w1, b1, w2, b2, gw, gb, eps=1e-5):
    xf = x.float()
    return F.layer_norm(xf + F.linear(torch.relu(F.linear(xf, w1.float(), b1)), w2.float(), b2), (D,), gw, gb, eps)


@pytest.mark.parametrize("T,Fh", [(1, 32), (63, 32), (64, 64), (255, 96), (256, 2048), (257, 1024), (1100 * 2, 2048), (5000, 4096)])
def test_forward_matches_fp32_definition(T, Fh):
    x, w1, b1, w2, b2, gw, gb = make(T, Fh, seed=T + Fh)
    out = ffn_forward_bf16(x, w1, b1, pack_w2_bf16(w2), b2, gw, gb)
    ref = ref_fp32(x, w1, b1, w2, b2, gw, gb)
    err = (out.float() - ref).abs()
    assert torch.isfinite(out.float()).all()
    assert float((err / ref.abs().clamp(min=1.0)).max()) < 2 ** -7, (T, Fh)
    assert float(err.mean()) < 3e-3


def test_rows_are_independent_and_tail_is_masked():
    """A token's output depends on its own row only; rows past `tokens` are never written."""
    x, w1, b1, w2, b2, gw, gb = make(300, 64, seed=5)
    w2p = pack_w2_bf16(w2)
    full = ffn_forward_bf16(x, w1, b1, w2p, b2, gw, gb)
    part = ffn_forward_bf16(x[:129].contiguous(), w1, b1, w2p, b2, gw, gb)
    assert torch.equal(full[:129], part)
    lib = _lib.load()
    out = torch.full((300, D), 7.0, device="cuda", dtype=torch.bfloat16)
    _lib.check(lib.msda_ffn_forward_bf16(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2p.data_ptr(), b2.data_ptr(), gw.data_ptr(),
                                         gb.data_ptr(), 1e-5, 129, D, 64, out.data_ptr(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert torch.equal(out[:129], part) and bool((out[129:] == 7.0).all())


def test_weight_packing_is_the_documented_permutation():
    """inside every 32 hidden columns, position 8 q + j holds column 4 q + j (j < 4) or 16 + 4 q + (j - 4)"""
    w2 = torch.arange(D * 64, device="cuda", dtype=torch.float32).remainder(251).to(torch.bfloat16).view(D, 64)
    perm = [4 * (p >> 3) + (p & 7) if (p & 7) < 4 else 16 + 4 * (p >> 3) + (p & 7) - 4 for p in range(32)]
    assert sorted(perm) == list(range(32))
    idx = torch.tensor([32 * g + perm[p] for g in range(2) for p in range(32)], device="cuda")
    assert torch.equal(pack_w2_bf16(w2), w2[:, idx])


def test_bad_arguments_are_rejected():
    lib = _lib.load()
    x, w1, b1, w2, b2, gw, gb = make(8, 32)
    out = torch.empty_like(x)
    s = torch.cuda.current_stream().cuda_stream
    args = lambda dm, df: (x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), gw.data_ptr(), gb.data_ptr(), 1e-5,
                           8, dm, df, out.data_ptr(), s)
    assert lib.msda_ffn_forward_bf16(*args(128, 32)) == -2      # d_model must be 256
    assert lib.msda_ffn_forward_bf16(*args(256, 48)) == -2      # d_ffn % 32
    assert lib.msda_ffn_forward_bf16(x.data_ptr(), None, b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), gw.data_ptr(), gb.data_ptr(),
                                     1e-5, 8, 256, 32, out.data_ptr(), s) == -1
    with pytest.raises(RuntimeError):
        ffn_forward_bf16(x.cpu(), w1.cpu(), b1.cpu(), w2.cpu(), b2.cpu(), gw.cpu(), gb.cpu())


def test_function_gradients_match_fp32_autograd():
    """FusedFFNFunction (fused forward, recomputing GEMM backward in bf16) against fp32 autograd of the op-by-op block."""
    x, w1, b1, w2, b2, gw, gb = make(512, 256, seed=3)
    leaves = [t.clone().requires_grad_(True) for t in (x, w1, b1, w2, b2, gw, gb)]
    out = FusedFFNFunction.apply(*leaves, 1e-5)
    go = torch.randn(512, D, device="cuda", generator=torch.Generator(device="cuda").manual_seed(9)).to(torch.bfloat16)
    out.backward(go)
    refl = [t.float().clone().requires_grad_(True) for t in (x, w1, b1, w2, b2, gw, gb)]
    ref_fp32(*refl).backward(go.float())
    for a, b, name in zip(leaves, refl, ("x", "w1", "b1", "w2", "b2", "ln_w", "ln_b")):
        err = (a.grad.float() - b.grad).abs()   # bf16 GEMMs + bf16 storage of the gradients of the bf16 leaves
        assert float(err.max()) / (float(b.grad.abs().max()) + 1e-12) < 0.15, name   # (single elements: a few bf16 ulps)
        assert float(err.mean()) / (float(b.grad.abs().mean()) + 1e-12) < 2e-2, name


def test_cached_function_equals_the_uncached_one_and_keeps_fp32_weight_gradients():
    """FusedFFNCachedFunction (packed parameters kept by the caller, float32 master parameters handed over) against FusedFFNFunction on
    the bf16 casts: the same kernels -> equal output and equal input gradient; weight gradients equal before the uncached form's bf16
    rounding, and closer to fp32 autograd than it"""
    from richsem_amd.functions.ffn import FusedFFNCachedFunction, pack_ffn
    x, w1, b1, w2, b2, gw, gb = make(4608, 256, seed=5)
    go = torch.randn(4608, D, device="cuda", generator=torch.Generator(device="cuda").manual_seed(9)).to(torch.bfloat16)
    masters = [t.float().clone().requires_grad_(True) for t in (w1, b1, w2, b2, gw, gb)]
    xa = x.clone().requires_grad_(True)
    pk = pack_ffn(masters[0], masters[1], masters[2])
    out_c = FusedFFNCachedFunction.apply(xa, pk, 1e-5, *masters)
    out_c.backward(go)
    leaves = [t.clone().requires_grad_(True) for t in (x, w1, b1, w2, b2, gw, gb)]
    out_u = FusedFFNFunction.apply(*leaves, 1e-5)
    out_u.backward(go)
    assert torch.equal(out_c, out_u) and torch.equal(xa.grad, leaves[0].grad)
    refl = [t.float().clone().requires_grad_(True) for t in (x, w1, b1, w2, b2, gw, gb)]
    ref_fp32(*refl).backward(go.float())
    for i, name in ((0, "w1"), (2, "w2")):
        gc, gu, gr = masters[i].grad, leaves[1 + i].grad, refl[1 + i].grad
        assert gc.dtype == torch.float32 and gu.dtype == torch.bfloat16
        assert torch.equal(gc.to(torch.bfloat16), gu), name                                  # the uncached form = this one, rounded
        ec, eu = float((gc - gr).abs().mean()), float((gu.float() - gr).abs().mean())
        assert ec <= eu * 1.001, (name, ec, eu)
    for i in (1, 3, 4, 5):      # (bias / LayerNorm gradients: token sums whose order of accumulation is not fixed)
        a, b = masters[i].grad, leaves[1 + i].grad.float()
        assert float((a - b).abs().max()) <= 1e-3 * float(b.abs().max()), i


def test_module_fused_path_equals_op_by_op_path_within_bf16():
    torch.manual_seed(0)
    m = FFN(256, 2048, dropout=0.0).cuda()
    src = torch.randn(2, 700, 256, device="cuda").to(torch.bfloat16)
    assert not m._can_fuse(src)          # 1400 tokens: below the size from which the one-kernel block pays
    m.fused_min_tokens = 0
    assert m._can_fuse(src)
    fused = m(src)
    m.fused = False
    plain = m.to(torch.bfloat16)(src)
    assert fused.shape == src.shape and fused.dtype == torch.bfloat16
    assert float((fused.float() - plain.float()).abs().mean()) < 6e-3


@pytest.mark.parametrize("tokens,cout,cin", [(5000, 256, 256), (4097, 384, 256), (4500, 256, 2048), (300, 128, 128)])
def test_linear_weight_gradient_kernel(tokens, cout, cin):
    """functions/linear.py: dW = dY^T X through the convolution weight-gradient kernel against fp32 on the same bf16 operands"""
    from richsem_amd.functions.linear import linear_bf16, linear_wgrad_bf16
    torch.manual_seed(tokens)
    dy = torch.randn(tokens, cout, device="cuda").to(torch.bfloat16)
    x = torch.randn(tokens, cin, device="cuda").to(torch.bfloat16)
    want = dy.float().t() @ x.float()
    got = linear_wgrad_bf16(dy, x)
    assert got.dtype == torch.float32 and float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())
    # the autograd function around it: gradients in the parameters' dtype, equal to the op-by-op bf16 linear's within bf16
    w = (torch.randn(cout, cin, device="cuda") * cin ** -0.5).requires_grad_(True)
    b = torch.zeros(cout, device="cuda", requires_grad=True)
    xin = x.clone().requires_grad_(True)
    linear_bf16(xin, w, b).backward(dy)
    wr, br, xr = w.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True), x.float().requires_grad_(True)
    torch.nn.functional.linear(xr, wr.to(torch.bfloat16).float(), br).backward(dy.float())
    assert w.grad.dtype == torch.float32 and b.grad.dtype == torch.float32 and xin.grad.dtype == torch.bfloat16
    for got, ref in ((w.grad, wr.grad), (b.grad, br.grad), (xin.grad.float(), xr.grad)):
        assert float((got - ref).abs().max()) <= 1e-2 * float(ref.abs().max())


@pytest.mark.parametrize("tokens,n", [(1000, 2048), (193, 64), (47, 128), (4097, 1024)])
def test_lin256_kernel(tokens, n):
    """csrc/lin256_mfma.hip: relu(x W^T + b), x W^T + b and (x W^T) * (mask > 0) against fp32 on the same bf16 operands (one bf16
    rounding of the result)"""
    from richsem_amd.functions.linear import lin256, lin256_pack
    torch.manual_seed(n + tokens)
    x = torch.randn(tokens, 256, device="cuda").to(torch.bfloat16)
    w = (torch.randn(n, 256, device="cuda") / 16).to(torch.bfloat16)
    b = torch.randn(n, device="cuda")
    wp = lin256_pack(w)
    lin = x.float() @ w.float().t()
    for got, want in ((lin256(x, wp, b, relu=True), torch.relu(lin + b)), (lin256(x, wp, b), lin + b), (lin256(x, wp), lin)):
        assert got.dtype == torch.bfloat16 and float((got.float() - want).abs().max()) <= 2 ** -7 * float(want.abs().max())
    h = torch.relu(lin + b).to(torch.bfloat16)
    got = lin256(x, wp, relu_mask=h)
    want = lin * (h > 0)
    assert float((got.float() - want).abs().max()) <= 2 ** -7 * float(want.abs().max())
    assert bool(((got == 0) | (h > 0)).all())


@pytest.mark.parametrize("tokens,n", [(1000, 256), (4097, 384), (50, 32)])
def test_lin256_f32_split_kernel(tokens, n):
    """fp32 in / out on bf16 MFMAs with hi + lo split operands: fp32-level accuracy (1e-5 of the output range against fp64)"""
    from richsem_amd.functions.linear import lin256_f32, lin256_f32_pack
    torch.manual_seed(n + tokens)
    x = torch.randn(tokens, 256, device="cuda")
    w = torch.randn(n, 256, device="cuda") / 16
    b = torch.randn(n, device="cuda")
    want = x.double() @ w.double().t() + b.double()
    got = lin256_f32(x, lin256_f32_pack(w), n, b)
    assert got.dtype == torch.float32 and float((got.double() - want).abs().max()) <= 1e-5 * float(want.abs().max())
    got0 = lin256_f32(x, lin256_f32_pack(w), n)
    assert float((got0.double() - (want - b.double())).abs().max()) <= 1e-5 * float(want.abs().max())


@pytest.mark.parametrize("tokens", [1, 1000, 4097])
def test_add_layernorm_function(tokens):
    """AddLayerNormFunction (residual add + LayerNorm, one kernel each way) against fp32 autograd of F.layer_norm(a + b) on the same bf16
    inputs: output within one bf16 rounding, gradients at bf16 level"""
    import torch.nn.functional as F
    from richsem_amd.functions import AddLayerNormFunction
    torch.manual_seed(tokens)
    a = torch.randn(tokens, 256, device="cuda").to(torch.bfloat16)
    b = (0.5 * torch.randn(tokens, 256, device="cuda")).to(torch.bfloat16)
    w, bi = torch.rand(256, device="cuda") + 0.5, torch.randn(256, device="cuda")
    go = torch.randn(tokens, 256, device="cuda").to(torch.bfloat16)
    a1, b1, w1, bi1 = (t.clone().requires_grad_(True) for t in (a, b, w, bi))
    out = AddLayerNormFunction.apply(a1, b1, w1, bi1, 1e-5)
    out.backward(go)
    a2, b2, w2, bi2 = (t.float().clone().requires_grad_(True) for t in (a, b, w, bi))
    ref = F.layer_norm(a2 + b2, (256,), w2, bi2, 1e-5)
    ref.backward(go.float())
    assert float((out.float() - ref).abs().max()) <= 2 ** -7 * float(ref.abs().max())
    for got, want in ((a1.grad.float(), a2.grad), (b1.grad.float(), b2.grad), (w1.grad, w2.grad), (bi1.grad, bi2.grad)):
        assert float((got - want).abs().max()) <= 2e-2 * float(want.abs().max()) + 1e-3
