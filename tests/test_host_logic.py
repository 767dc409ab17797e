"""CPU: host-side logic of the drop-in (argument checks of the shim, workload geometry, byte counts)."""
import sys

import pytest
import torch

import richsem_amd
from richsem_amd import workload as W
from richsem_amd import MultiScaleDeformableAttention as MSDA
from richsem_amd.functions import MSDeformAttnFunction
from richsem_amd.modules import MSDeformAttn


def small_inputs(dtype=torch.float32):
    call = W.Call("t", 1, 2, 4, 2, [(6, 4), (3, 2)], 5, False)
    return call, W.make_inputs(call, "uniform", seed=1, dtype=dtype)


def test_cpu_tensors_are_rejected_like_the_reference():
    """reference src/ms_deform_attn.h:38,60: AT_ERROR("Not implemented on the CPU") -- and no silent fallback."""
    _, t = small_inputs()
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        MSDA.ms_deform_attn_forward(t["value"], t["shapes"], t["lsi"], t["loc"], t["aw"], 64)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        MSDA.ms_deform_attn_backward(t["value"], t["shapes"], t["lsi"], t["loc"], t["aw"], t["grad_out"], 64)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        MSDeformAttnFunction.apply(t["value"], t["shapes"], t["lsi"], t["loc"], t["aw"], 64)


def test_dropin_module_name():
    shim = richsem_amd.install_dropin()
    import MultiScaleDeformableAttention as ref_name
    assert hasattr(ref_name, "ms_deform_attn_forward") and hasattr(ref_name, "ms_deform_attn_backward")
    assert sys.modules["MultiScaleDeformableAttention"] is not None
    assert shim.ms_deform_attn_forward is MSDA.ms_deform_attn_forward


def test_module_state_dict_keys_and_init_match_reference():
    """reference ops/modules/ms_deform_attn.py:50-76: parameter names and the initial sampling pattern."""
    m = MSDeformAttn(256, 4, 8, 4)
    assert sorted(m.state_dict().keys()) == sorted(
        f"{n}.{p}" for n in ("sampling_offsets", "attention_weights", "value_proj", "output_proj")
        for p in ("weight", "bias"))
    assert m.im2col_step == 64
    assert m.sampling_offsets.weight.abs().max() == 0 and m.attention_weights.weight.abs().max() == 0
    assert m.attention_weights.bias.abs().max() == 0 and m.value_proj.bias.abs().max() == 0
    b = m.sampling_offsets.bias.detach().view(8, 4, 4, 2)
    # head 0 points along +x, head 2 along +y; point k is k+1 steps out; same for every level
    assert torch.allclose(b[0, :, :, 0], torch.tensor([1., 2., 3., 4.]).expand(4, 4))
    assert torch.allclose(b[0, :, :, 1], torch.zeros(4, 4), atol=1e-6)
    assert torch.allclose(b[2, :, :, 1], torch.tensor([1., 2., 3., 4.]).expand(4, 4))
    assert torch.allclose(b[1, 0, 0], torch.tensor([1., 1.]), atol=1e-6)      # diagonal, max-norm 1
    assert torch.equal(b[:, 0], b[:, 3])
    with pytest.raises(ValueError):
        MSDeformAttn(250, 4, 8, 4)


def test_module_rejects_bad_reference_points_before_the_op():
    m = MSDeformAttn(32, 2, 4, 2)
    q = torch.zeros(1, 5, 32)
    src = torch.zeros(1, 30, 32)
    shapes = torch.tensor([[6, 4], [3, 2]])
    lsi = torch.tensor([0, 24])
    with pytest.raises(ValueError, match="Last dim of reference_points must be 2 or 4"):
        m(q, torch.zeros(1, 5, 2, 3), src, shapes, lsi)


def test_pyramid_shapes_and_algorithmic_bytes():
    """SURVEY.md section 8 / BASELINE.md section 3 figures."""
    E, Dd, Em = W.call_E(), W.call_Dd(), W.call_Em()
    assert E.shapes == [(100, 168), (50, 84), (25, 42), (13, 21)] and E.S == 22323 and E.Lq == 22323
    assert Em.shapes == [(160, 160), (80, 80), (40, 40), (20, 20)] and Em.S == 34000
    assert (E.bytes_fwd(), E.bytes_bwd()) == (160011264, 274305024)
    assert (Dd.bytes_fwd(), Dd.bytes_bwd()) == (51308544, 100380672)
    step = sum(r * (c.bytes_fwd() + c.bytes_bwd()) for c, r in W.training_step_calls())
    assert abs(step - 3.52e9) < 0.01e9
    sh, lsi = W.level_tensors(E)
    assert lsi.tolist() == [0, 16800, 21000, 22050]


def test_make_inputs_is_seeded_and_well_formed():
    call = W.shrunk(W.call_E(1), 8)
    a = W.make_inputs(call, "init", seed=3)
    b = W.make_inputs(call, "init", seed=3)
    c = W.make_inputs(call, "init", seed=4)
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert not torch.equal(a["loc"], c["loc"])
    assert a["loc"].shape == (1, call.Lq, 8, 4, 4, 2) and a["loc"].is_contiguous()
    assert torch.allclose(a["aw"].sum((-1, -2)), torch.ones(1, call.Lq, 8), atol=1e-5)
    u = W.make_inputs(call, "uniform", seed=3)
    assert 0 <= float(u["loc"].min()) and float(u["loc"].max()) < 1


def test_tiled_plan_geometry():
    """Host-only launch plan of the LDS-window kernels (msda_tiled_plan): applicability and LDS budget."""
    from richsem_amd import _lib
    E = W.call_E()
    sh, lsi = W.level_tensors(E)
    p = _lib.tiled_plan(E.N, E.S, E.M, E.D, E.L, E.Lq, E.P, sh.tolist(), lsi.tolist())
    assert p["applicable"] == 1
    assert p["GY"] * p["GX"] >= 24 and p["max_region_queries"] <= 512
    assert p["lds_bytes"] <= 160 * 1024 and p["phases"] >= 1
    assert p["grid"] == 8 * 2 * p["GY"] * p["GX"] * 2      # 16 (image, head) pairs over 8 XCDs, two channel halves
    assert p["lds_bytes"] <= 80 * 1024                     # two gather workgroups share a CU
    Em = W.call_Em()
    sh, lsi = W.level_tensors(Em)
    assert _lib.tiled_plan(Em.N, Em.S, Em.M, Em.D, Em.L, Em.Lq, Em.P, sh.tolist(), lsi.tolist())["applicable"] == 1
    # decoder-shaped (Lq != S), other channel counts, or levels not laid out back to back: direct kernels
    Dd = W.call_Dd()
    sh, lsi = W.level_tensors(Dd)
    assert _lib.tiled_plan(Dd.N, Dd.S, Dd.M, Dd.D, Dd.L, Dd.Lq, Dd.P, sh.tolist(), lsi.tolist())["applicable"] == 0
    sh, lsi = W.level_tensors(E)
    assert _lib.tiled_plan(E.N, E.S, E.M, 64, E.L, E.Lq, E.P, sh.tolist(), lsi.tolist())["applicable"] == 0
    bad = lsi.tolist()
    bad[1], bad[2] = bad[2], bad[1]
    assert _lib.tiled_plan(E.N, E.S, E.M, E.D, E.L, E.Lq, E.P, sh.tolist(), bad)["applicable"] == 0


def test_region_partition_covers_every_query_once():
    """The region <-> query assignment of the LDS-window kernels (richsem_amd/csrc/msda_tiled.h, region_first):
    pixel row r of a level with H rows belongs to region row g iff (2r+1)*G // (2H) == g."""
    def region_first(H, g, G):
        return (2 * H * g + G - 1) // (2 * G)
    for H in (1, 2, 3, 13, 21, 25, 42, 50, 84, 100, 168, 160):
        for G in (1, 2, 7, 11, 13, 21):
            owner = [(2 * r + 1) * G // (2 * H) for r in range(H)]
            for g in range(G):
                lo, hi = region_first(H, g, G), region_first(H, g + 1, G)
                assert [r for r in range(H) if owner[r] == g] == list(range(lo, hi))
            assert region_first(H, 0, G) == 0 and region_first(H, G, G) == H


def test_levelsum_plan_hands_over_all_levels_of_decoder_calls():
    """Host-only plan of the level-sum backward kernel (msda_levelsum_plan): decoder-shaped calls hand over every level
    (the finest one in row bands), encoder-sized calls too, and a level too large for eight bands stays with the
    direct kernel's atomics while the small ones are still taken."""
    from richsem_amd import _lib

    def plan(call):
        sh, lsi = W.level_tensors(call)
        return _lib.levelsum_plan(call.N, call.S, call.M, call.D, call.L, call.Lq, call.P, sh.tolist(), lsi.tolist())

    Dd = W.call_Dd(2)
    p = plan(Dd)
    assert p["levels_mask"] == 0b1111
    assert p["windows"] == 4 + 1 + 1 + 1            # 100 x 168 in four bands of 25 rows; the other levels whole
    assert p["max_rows"] == 50                       # the 50 x 84 level, whole
    assert p["slices"] == 8 and p["grid"] == 16 * 7 * 8
    assert p["lds_bytes"] <= 150 * 1024 and p["lds_bytes"] == 25 * 168 * 4 * 8

    p = plan(W.call_E(2))                            # the direct backward of an encoder-sized call (scattered data)
    assert p["levels_mask"] == 0b1111 and p["windows"] == 7
    huge = W.Call("huge", 1, 1, 32, 4, [(20, 20)], 300000, False)
    assert plan(huge)["levels_mask"] == 0            # Lq * P beyond what one workgroup should walk: row atomics

    big = W.Call("big", 1, 1, 32, 4, [(300, 300), (9, 9)], 64, False)
    p = plan(big)
    assert p["levels_mask"] == 0b10 and p["windows"] == 1   # 9 x 9 receives 256 points (>= 2 per pixel); 300 x 300 does not fit

    odd = W.Call("odd", 1, 2, 30, 2, [(91, 70), (5, 5)], 500, False)
    p = plan(odd)
    assert p["levels_mask"] == 0b11 and p["windows"] == 3 and p["max_rows"] == 46 and p["slices"] == 8


def test_dn_group_count_matches_the_restated_reference_arithmetic():
    """richsem_amd/dn.py (host ints) against oracle/dn_oracle.py for the denoising-group arithmetic (dn_components.py:27-41),
    and the oracle's mask against the properties the reference's comments state."""
    from oracle import dn_oracle
    from richsem_amd.dn import dn_group_count
    for known in ([12, 12], [3, 0, 7], [0, 0], [1], [250, 3]):
        for dn in (0, 1, 3, 99, 100, 101, 1000):
            for add_gt in (False, True):
                want = dn_oracle.prepare_for_cdn_indices(known, dn, 5, True, add_gt)["num_dn_group"]
                assert dn_group_count(dn, known, add_gt) == want
    o = dn_oracle.prepare_for_cdn_indices([3, 2], 100, 7, True)
    m, pad, gp = o["attn_mask"], o["pad_size"], o["group_pad"]
    assert m.shape == (pad + 7, pad + 7)
    assert m[pad:, :pad].all() and not m[:, pad:].any()              # match queries cannot see the denoising part; everyone sees the matching part
    for i in range(pad):
        for j in range(pad):
            assert m[i, j] == (i // gp != j // gp)                   # denoising groups cannot see each other


def test_input_projection_carries_the_reference_names_when_nested():
    """richsem.py:295-310 keeps the projections as ``self.input_proj = nn.ModuleList([nn.Sequential(conv, GroupNorm), ...])``: nested
    under that attribute the trainable mirror must produce / accept ``input_proj.{l}.{0,1}.{weight,bias}`` with strict=True"""
    from richsem_amd.backbone import InputProjection

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.input_proj = InputProjection(in_channels=(16, 32, 64), hidden=32, num_levels=4, groups=4)

    m = Model()
    want = {f"input_proj.{l}.{i}.{n}" for l in range(4) for i in (0, 1) for n in ("weight", "bias")}
    assert set(m.state_dict().keys()) == want
    # a reference-named checkpoint (shapes of nn.Conv2d / nn.GroupNorm) loads with strict=True and lands in the parameters
    ref = {}
    for l, (cin, k) in enumerate(((16, 1), (32, 1), (64, 1), (64, 3))):
        ref[f"input_proj.{l}.0.weight"] = torch.full((32, cin, k, k), float(l + 1))
        ref[f"input_proj.{l}.0.bias"] = torch.full((32,), 0.5 * l)
        ref[f"input_proj.{l}.1.weight"] = torch.full((32,), 2.0 + l)
        ref[f"input_proj.{l}.1.bias"] = torch.full((32,), -1.0 * l)
    missing, unexpected = m.load_state_dict(ref, strict=True)
    assert not missing and not unexpected
    assert float(m.input_proj[3][0].weight[0, 0, 0, 0]) == 4.0 and float(m.input_proj[2][1].bias[0]) == -2.0
    # and as the root module the keys are the list's own
    assert set(InputProjection(in_channels=(16,), hidden=32, num_levels=2, groups=4).state_dict().keys()) == \
        {f"{l}.{i}.{n}" for l in range(2) for i in (0, 1) for n in ("weight", "bias")}


def test_fold_bn_matches_the_reference_frozen_batchnorm():
    """richsem_amd.conv.fold_bn (the scale / shift every backbone convolution's epilogue applies) against the reference's own
    FrozenBatchNorm2d.forward (tests/golden/make_golden_frozenbn.py -> frozenbn_fold.npz)"""
    import os
    import numpy as np
    from richsem_amd.conv import fold_bn
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frozenbn_fold.npz"))
    t = {k: torch.from_numpy(z[k]) for k in z.files}
    scale, shift = fold_bn(t["weight"], t["bias"], t["running_mean"], t["running_var"])
    y = t["x"] * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    assert torch.allclose(y, t["y"], rtol=1e-6, atol=1e-6)


def test_wgrad_group_protocol_is_enforced(monkeypatch):
    """functions/linear.py: a deferring function hands autograd an EMPTY weight gradient that the group's boundary fills later.  That is
    only correct for aliases made by that group's boundary, each feeding exactly one function -- both are checked, not assumed: raw
    parameters inside an active group are not deferred, and an alias that feeds two functions raises at the boundary (autograd has summed
    the unwritten tensors by then) instead of training on garbage.  (The launch itself is replaced by a fill: no GPU here.)"""
    from richsem_amd.functions.linear import WgradGroup, wgrad_boundary
    monkeypatch.setattr(WgradGroup, "_launch", staticmethod(lambda pend: [t[2].fill_(7.0) for t in pend]))
    seen = {}

    class Deferring(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.group = seen["group"] = WgradGroup.active_for((w,))
            return x * 2

        @staticmethod
        def backward(ctx, dy):
            dw = torch.empty(3)
            if ctx.group is not None:
                ctx.group.pending.append((None, None, dw, None))
            else:
                dw.fill_(7.0)
            return dy * 2, dw

    w, x = torch.zeros(3, requires_grad=True), torch.ones(3, requires_grad=True)
    group = WgradGroup()
    (alias,) = wgrad_boundary(group, w)
    with group:
        y = Deferring.apply(x, alias)
        assert seen["group"] is group                     # the boundary's alias: deferred
        y_raw = Deferring.apply(x, w)
        assert seen["group"] is None                      # the raw parameter inside the active group: computed on the spot
    (y.sum() + y_raw.sum()).backward()
    assert torch.equal(w.grad, torch.full((3,), 14.0))
    other = WgradGroup()
    wgrad_boundary(other, w)
    with group:
        Deferring.apply(x, alias)
        assert seen["group"] is group
    with other:
        Deferring.apply(x, alias)
        assert seen["group"] is None                      # another group's alias
    w.grad = None
    group = WgradGroup()
    (alias,) = wgrad_boundary(group, w)
    with group:
        y = Deferring.apply(x, alias) + Deferring.apply(x, alias)
    with pytest.raises(RuntimeError, match="fed more than one function"):
        y.sum().backward()
