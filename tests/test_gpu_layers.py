"""GPU (-m gpu): the transformer layers against fixtures generated from the REFERENCE's own classes
(tests/golden/make_golden_layers.py: DeformableTransformerEncoderLayer, DeformableTransformerDecoderLayer and TransformerDecoder of
models/richsem/deformable_transformer.py, fp64, the operator bound to the reference's pure-PyTorch core): outputs, input gradients and
every parameter gradient.  The bf16 paths (new capability, no reference) are held to the fp32 path of the same module -- which the
fixtures pin -- at bf16 tolerance; the attention kernels to the softmax definition in fp32."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _fix(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    t = {k: torch.from_numpy(z[k]).cuda() for k in z.files if not k.startswith("param.")}
    sd = {k[len("param."):]: torch.from_numpy(z[k]).cuda() for k in z.files if k.startswith("param.") and not k.endswith(".grad")}
    gr = {k[len("param."):-len(".grad")]: torch.from_numpy(z[k]).cuda() for k in z.files if k.startswith("param.") and k.endswith(".grad")}
    return t, sd, gr


def _rel(a, b):
    return float((a - b).abs().max()) / (float(b.abs().max()) + 1e-300)


def _ratio(x, y, tag=None):
    """mean |x - y| / mean |y|; RICHSEM_REPORT=1 prints it (the bounds below are 2 x what MI355X measures: profiles/r04_bf16_bounds.txt)"""
    r = float((x.double() - y.double()).abs().mean()) / (float(y.double().abs().mean()) + 1e-12)
    if tag and os.environ.get("RICHSEM_REPORT"):
        print(f"[measured] {tag} {r:.4g}", flush=True)
    return r


def _check_params(mod, gr, tol):
    got = dict(mod.named_parameters())
    assert set(got) == set(gr), set(got) ^ set(gr)
    for k, want in gr.items():
        assert got[k].grad is not None, k
        assert _rel(got[k].grad, want) < tol, (k, _rel(got[k].grad, want))


def test_encoder_layer_against_the_reference_class():
    from richsem_amd.modules import DeformableTransformerEncoderLayer, get_reference_points
    t, sd, gr = _fix("layer_encoder_f64")
    layer = DeformableTransformerEncoderLayer(64, 96, dropout=0.0, activation="relu", n_levels=4, n_heads=2, n_points=4).cuda().double()
    assert set(layer.state_dict()) == set(sd)
    layer.load_state_dict(sd, strict=True)
    ref = get_reference_points(t["shapes"].tolist(), t["valid_ratios"], "cuda")          # the encoder's own :512-525
    assert _rel(ref.double(), t["reference_points"]) < 1e-6
    src, pos = t["src"].clone().requires_grad_(True), t["pos"].clone().requires_grad_(True)
    out = layer(src, pos, t["reference_points"], t["shapes"], t["lsi"], t["mask"])
    assert _rel(out, t["out"]) < 1e-10
    out.backward(t["grad_out"])
    assert _rel(src.grad, t["grad_src"]) < 1e-9 and _rel(pos.grad, t["grad_pos"]) < 1e-9
    _check_params(layer, gr, 1e-9)


def test_decoder_layer_against_the_reference_class():
    from richsem_amd.modules import DeformableTransformerDecoderLayer
    t, sd, gr = _fix("layer_decoder_f64")
    layer = DeformableTransformerDecoderLayer(64, 96, dropout=0.0, activation="relu", n_levels=4, n_heads=2, n_points=4).cuda().double()
    assert set(layer.state_dict()) == set(sd)
    layer.load_state_dict(sd, strict=True)
    tgt, qpos, mem = (t[k].clone().requires_grad_(True) for k in ("tgt", "query_pos", "memory"))
    out = layer(tgt=tgt, tgt_query_pos=qpos, tgt_reference_points=t["reference_points"], memory=mem, memory_key_padding_mask=t["memory_mask"],
                memory_level_start_index=t["lsi"], memory_spatial_shapes=t["shapes"], self_attn_mask=t["attn_mask"])
    assert _rel(out, t["out"]) < 1e-10
    out.backward(t["grad_out"])
    assert _rel(tgt.grad, t["grad_tgt"]) < 1e-9 and _rel(qpos.grad, t["grad_query_pos"]) < 1e-9 and _rel(mem.grad, t["grad_memory"]) < 1e-9
    _check_params(layer, gr, 1e-9)


def test_decoder_stack_against_the_reference_class():
    """TransformerDecoder.forward: sine embedding, ref_point_head, three layers, iterative box refinement with detached boxes"""
    from richsem_amd.modules import MLP, DeformableTransformerDecoderLayer, TransformerDecoder
    t, sd, gr = _fix("decoder_stack_f64")
    layer = DeformableTransformerDecoderLayer(64, 96, dropout=0.0, activation="relu", n_levels=4, n_heads=2, n_points=4)
    dec = TransformerDecoder(layer, 3, torch.nn.LayerNorm(64), d_model=64, query_dim=4, num_feature_levels=4)
    dec.bbox_embed = torch.nn.ModuleList([MLP(64, 64, 4, 3) for _ in range(3)])
    dec = dec.cuda().double()
    assert set(dec.state_dict()) == set(sd)
    dec.load_state_dict(sd, strict=True)
    tgt, mem, refu = (t[k].clone().requires_grad_(True) for k in ("tgt", "memory", "refpoints_unsigmoid"))
    hs, refs = dec(tgt=tgt, memory=mem, tgt_mask=t["attn_mask"], memory_key_padding_mask=t["memory_mask"], refpoints_unsigmoid=refu,
                   level_start_index=t["lsi"], spatial_shapes=t["shapes"], valid_ratios=t["valid_ratios"])
    hs, refs = torch.stack(hs), torch.stack(refs)
    # (the sine embedding's frequencies 10000 ** (2 (i // 2) / 128) are float32 in the reference, utils.py:145-146; the fixture took that
    # power on the CPU, here it is the GPU's: one float32 ulp apart, ~1e-9 after the ref_point_head)
    assert _rel(hs, t["hs"]) < 2e-8 and _rel(refs, t["refs"]) < 2e-8
    ((hs * t["grad_hs"]).sum() + (refs * t["grad_refs"]).sum()).backward()
    assert _rel(tgt.grad, t["grad_tgt"]) < 1e-7 and _rel(mem.grad, t["grad_memory"]) < 1e-7 and _rel(refu.grad, t["grad_refpoints"]) < 1e-7
    _check_params(dec, gr, 1e-7)


# ---- d_model = 256, 8 heads: the bf16 kernels (K = 256 projections, 8 x 32-channel heads, fused feed-forward block) against fixtures of the
#      REFERENCE's classes at that width (tests/golden/make_golden_layers.py: layers_256; parameters rebuilt by tests/layer_params.py) ------
def _fix256(name):
    import layer_params as LP
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    t = {k: torch.from_numpy(z[k]).cuda() for k in z.files if not k.startswith(("pgrad.", "pscale."))}
    t = {k: (v.double() if v.dtype == torch.float32 else v) for k, v in t.items()}
    return t, {k: v.cuda() for k, v in LP.unpack_grads(z).items()}


def _mean_rel(a, b):
    return float((a.double() - b.double()).abs().mean()) / (float(b.double().abs().mean()) + 1e-300)


def _errs_params(mod, gr):
    got = dict(mod.named_parameters())
    assert set(gr) <= set(got)
    return {k: (_rel(got[k].grad.double(), want), _mean_rel(got[k].grad, want)) for k, want in gr.items()}


def _build256(kind, dtype):
    import layer_params as LP
    from richsem_amd.modules import MLP, DeformableTransformerDecoderLayer, DeformableTransformerEncoderLayer, TransformerDecoder
    C, F, L, H, P = LP.D_MODEL, LP.D_FFN, LP.LEVELS, LP.HEADS, LP.POINTS
    if kind == "encoder":
        m, seed = DeformableTransformerEncoderLayer(C, F, dropout=0.0, activation="relu", n_levels=L, n_heads=H, n_points=P), 400
    elif kind == "decoder":
        m, seed = DeformableTransformerDecoderLayer(C, F, dropout=0.0, activation="relu", n_levels=L, n_heads=H, n_points=P), 410
    else:
        layer = DeformableTransformerDecoderLayer(C, F, dropout=0.0, activation="relu", n_levels=L, n_heads=H, n_points=P)
        m, seed = TransformerDecoder(layer, 2, torch.nn.LayerNorm(C), d_model=C, query_dim=4, num_feature_levels=L), 420
        m.bbox_embed = torch.nn.ModuleList([MLP(C, C, 4, 3) for _ in range(2)])
    return LP.fill(m.double(), seed).to(dtype).cuda()


def _run256(kind, dtype, act):
    """the mirror of `kind` with parameters in `dtype` on activations cast to `act` -> (outputs, input gradients, module)"""
    t, gr = _fix256({"encoder": "layer256_encoder_f64", "decoder": "layer256_decoder_f64", "stack": "decoder256_stack_f64"}[kind])
    m = _build256(kind, dtype)
    if kind == "encoder":
        m.fused_min_tokens = 0                       # (256 tokens: the fused feed-forward kernel is exercised, whatever its speed here)
    leaf = lambda k: t[k].to(act).clone().requires_grad_(True)
    if kind == "encoder":
        src, pos = leaf("src"), leaf("pos")
        out = m(src, pos, t["reference_points"].to(torch.float32 if act == torch.bfloat16 else act), t["shapes"], t["lsi"], t["mask"])
        out.backward(t["grad_out"].to(out.dtype))
        return {"out": (out, t["out"]), "grad_src": (src.grad, t["grad_src"]), "grad_pos": (pos.grad, t["grad_pos"])}, m, gr
    if kind == "decoder":
        tgt, qpos, mem = leaf("tgt"), leaf("query_pos"), leaf("memory")
        out = m(tgt=tgt, tgt_query_pos=qpos, tgt_reference_points=t["reference_points"].to(torch.float32 if act == torch.bfloat16 else act),
                memory=mem, memory_key_padding_mask=t["memory_mask"], memory_level_start_index=t["lsi"], memory_spatial_shapes=t["shapes"],
                self_attn_mask=t["attn_mask"])
        out.backward(t["grad_out"].to(out.dtype))
        return {"out": (out, t["out"]), "grad_tgt": (tgt.grad, t["grad_tgt"]), "grad_query_pos": (qpos.grad, t["grad_query_pos"]),
                "grad_memory": (mem.grad, t["grad_memory"])}, m, gr
    tgt, mem = leaf("tgt"), leaf("memory")
    refu = t["refpoints_unsigmoid"].to(torch.float32 if act == torch.bfloat16 else act).clone().requires_grad_(True)
    hs, refs = m(tgt=tgt, memory=mem, tgt_mask=t["attn_mask"], memory_key_padding_mask=t["memory_mask"], refpoints_unsigmoid=refu,
                 level_start_index=t["lsi"], spatial_shapes=t["shapes"], valid_ratios=t["valid_ratios"].to(refu.dtype))
    hs, refs = torch.stack(hs), torch.stack(refs)
    ((hs.double() * t["grad_hs"]).sum() + (refs.double() * t["grad_refs"]).sum()).backward()
    return {"hs": (hs, t["hs"]), "refs": (refs, t["refs"]), "grad_tgt": (tgt.grad, t["grad_tgt"]), "grad_memory": (mem.grad, t["grad_memory"]),
            "grad_refpoints": (refu.grad, t["grad_refpoints"])}, m, gr


@pytest.mark.parametrize("kind", ["encoder", "decoder", "stack"])
def test_layers256_fp64_mirrors_equal_the_reference_classes(kind):
    """the fixture and the rebuilt parameters belong together: the fp64 mirrors reproduce the reference's outputs to what the fixture's
    storage keeps (float32 activations: 6e-8; float16 gradient mantissas: 5e-4)"""
    res, m, gr = _run256(kind, torch.float64, torch.float64)
    for k, (got, want) in res.items():
        assert _rel(got.double(), want) < 5e-7, (k, _rel(got.double(), want))
    for k, (mx, _) in _errs_params(m, gr).items():
        assert mx < 1e-3, (k, mx)


# bf16 activations / fp32 master parameters on the library's kernels against the reference's fp64 outputs: (max error / max |want|,
# mean error / mean |want|) per tensor.  The bounds are 2 x what MI355X measures (tools/measure_layer256.py, profiles/r04_layer256.txt);
# PyTorch's own bf16 ops on the same fixtures measure the same or worse on every tensor (same file: e.g. linear1.weight 0.21 for both --
# a hidden unit whose sign flips under bf16 rounding moves its whole row of the gradient -- sampling_offsets.weight 0.10 here, 0.30 there)
BF16_BOUNDS = {
    "encoder": {"out": (1.1e-2, 7e-3), "grad_src": (0.14, 3.6e-2), "grad_pos": (0.13, 4.5e-2), "params": (0.43, 6.2e-2)},
    "decoder": {"out": (1.5e-2, 1e-2), "grad_tgt": (0.16, 5.7e-2), "grad_query_pos": (0.1, 5.9e-2), "grad_memory": (7e-2, 6e-2), "params": (0.42, 8e-2)},
    "stack": {"hs": (1.5e-2, 1.1e-2), "refs": (6.3e-3, 1.6e-3), "grad_tgt": (0.11, 6.4e-2), "grad_memory": (6e-2, 7e-2), "grad_refpoints": (0.17, 6.5e-2),
              "params": (0.41, 0.137)},
}


@pytest.mark.parametrize("kind", ["encoder", "decoder", "stack"])
def test_layers256_bf16_kernels_against_the_reference_classes(kind):
    res, m, gr = _run256(kind, torch.float32, torch.bfloat16)
    bounds = BF16_BOUNDS[kind]
    for k, (got, want) in res.items():
        assert got.dtype in (torch.bfloat16, torch.float32)
        e = (_rel(got.double(), want), _mean_rel(got, want))
        assert e[0] < bounds[k][0] and e[1] < bounds[k][1], (k, e)
    worst = {k: e for k, e in _errs_params(m, gr).items() if e[0] > bounds["params"][0] or e[1] > bounds["params"][1]}
    assert not worst, worst


# ---- the attention kernels (csrc/attn_mfma.hip) against the definition -------------------------------------------------------------
def _attention_reference(qk, v, mask, heads):
    nq, bs, c2 = qk.shape
    C = c2 // 2
    q, k = qk[..., :C].float(), qk[..., C:].float()
    split = lambda x: x.reshape(nq, bs, heads, 32).permute(1, 2, 0, 3)          # (bs, heads, nq, 32)
    s = split(q) @ split(k).transpose(-1, -2) / 32 ** 0.5
    if mask is not None:
        s = s.masked_fill(mask[None, None], float("-inf"))
    return (torch.softmax(s, -1) @ split(v.float())).permute(2, 0, 1, 3).reshape(nq, bs, C)


@pytest.mark.parametrize("batch_first", [False, True])
@pytest.mark.parametrize("nq,bs,heads,masked", [(1092, 2, 8, True), (37, 1, 2, True), (64, 3, 1, False), (100, 2, 4, True), (1, 1, 1, False),
                                                 (900, 2, 8, False)])
def test_attention_kernels_against_the_definition(nq, bs, heads, masked, batch_first):
    from richsem_amd.functions.attention import masked_self_attention
    g = torch.Generator(device="cuda").manual_seed(nq * 7 + heads)
    C = heads * 32
    qk = (torch.randn(nq, bs, 2 * C, device="cuda", generator=g) * 1.5).to(torch.bfloat16)
    v = torch.randn(nq, bs, C, device="cuda", generator=g).to(torch.bfloat16)
    mask = None
    if masked:          # a denoising-style block mask plus random holes; every query keeps itself
        mask = torch.rand(nq, nq, device="cuda", generator=g) < 0.3
        mask[nq // 3:, : nq // 3] = True
        mask.fill_diagonal_(False)
    go = torch.randn(nq, bs, C, device="cuda", generator=g).to(torch.bfloat16)
    a, b = qk.clone().requires_grad_(True), v.clone().requires_grad_(True)
    if batch_first:      # the same tokens, stored (bs, nq, C)
        a_, b_ = a.transpose(0, 1).contiguous(), b.transpose(0, 1).contiguous()
        out = masked_self_attention(a_, b_, mask, heads, batch_first=True).transpose(0, 1)
    else:
        out = masked_self_attention(a, b, mask, heads)
    out.backward(go)
    ar, br = qk.float().requires_grad_(True), v.float().requires_grad_(True)
    want = _attention_reference(ar, br, mask, heads)
    want.backward(go.float())
    scale = lambda x: float(x.abs().max()) + 1e-12
    assert out.dtype == torch.bfloat16 and float((out.float() - want).abs().max()) < 2e-2 * scale(want)
    assert float((a.grad.float() - ar.grad).abs().max()) < 3e-2 * scale(ar.grad), float((a.grad.float() - ar.grad).abs().max()) / scale(ar.grad)
    assert float((b.grad.float() - br.grad).abs().max()) < 3e-2 * scale(br.grad)
    assert float((a.grad.float() - ar.grad).abs().mean()) < 4e-3 * scale(ar.grad)


def test_mask_bits_of_a_recycled_address_are_not_stale():
    """Two denoising layouts of equal shape and different group_pad, one after the other with the first mask freed in between (round-3
    advisor finding): ``prepare_dn_layout`` fills ``torch.empty`` buffers with a raw kernel (``_version`` 0) and the caching allocator
    would hand the second mask the first one's address -- the bit cache must not answer with the first mask's bits."""
    from richsem_amd import dn
    from richsem_amd.functions.attention import mask_bits, masked_self_attention
    heads, bs = 2, 1
    outs, masks = [], []
    g = torch.Generator(device="cuda").manual_seed(11)
    C = heads * 32
    lay = dn.prepare_dn_layout([4], dn_number=100, num_queries=40)        # pad_size 200, group_pad 8
    nq = lay["attn_mask"].shape[0]
    qk = (torch.randn(nq, bs, 2 * C, device="cuda", generator=g) * 1.5).to(torch.bfloat16)
    v = torch.randn(nq, bs, C, device="cuda", generator=g).to(torch.bfloat16)
    del lay
    for max_gt in (4, 5):          # both give pad_size 200 (dn_components.py:27-39, 65-67): 25 groups of 2 x 4, 20 groups of 2 x 5
        lay = dn.prepare_dn_layout([max_gt], dn_number=100, num_queries=40)
        mask = lay["attn_mask"]
        assert mask.shape == (nq, nq)
        bits = mask_bits(mask)[0]
        want_bits = mask_bits(mask.clone())[0]          # (a different address: always rebuilt)
        assert torch.equal(bits, want_bits)
        mask_bits(mask)                                   # the cache entry is this mask's again, as in a training step
        outs.append(masked_self_attention(qk, v, mask, heads).float())
        want = _attention_reference(qk, v, mask, heads)
        assert float((outs[-1] - want).abs().max()) < 2e-2 * float(want.abs().max())
        masks.append(mask.clone())
        del lay, mask, bits                               # the cache holds the tensor: its address must not be handed out again
    assert not torch.equal(masks[0], masks[1]) and float((outs[0] - outs[1]).abs().max()) > 1e-3


def test_lin256_row_mask_and_stacked_projection():
    from richsem_amd.functions.linear import StackedValueProjFunction, Lin256Function, pack_linear256
    torch.manual_seed(3)
    T = 1000
    x = torch.randn(2, T // 2, 256, device="cuda").to(torch.bfloat16).requires_grad_(True)
    ws = [torch.nn.Parameter(torch.randn(256, 256, device="cuda") * 0.05) for _ in range(3)]
    bs_ = [torch.nn.Parameter(torch.randn(256, device="cuda") * 0.1) for _ in range(3)]
    mask = torch.zeros(2, T // 2, dtype=torch.bool, device="cuda")
    mask[1, 7:40] = True
    outs = StackedValueProjFunction.apply(x, pack_linear256(ws, bs_), mask, *ws, *bs_)
    gos = [torch.randn(2, T // 2, 256, device="cuda").to(torch.bfloat16) for _ in range(3)]
    sum((o.float() * g.float()).sum() for o, g in zip(outs, gos)).backward()
    xr = x.detach().float().requires_grad_(True)
    wr = [w.detach().to(torch.bfloat16).float().requires_grad_(True) for w in ws]
    br = [b.detach().clone().requires_grad_(True) for b in bs_]
    want = [torch.nn.functional.linear(xr, w, b).masked_fill(mask[..., None], 0.0) for w, b in zip(wr, br)]
    sum((o * g.float()).sum() for o, g in zip(want, gos)).backward()
    for o, w_ in zip(outs, want):
        assert o.is_contiguous() and float((o.float() - w_).abs().max()) < 2e-2 * float(w_.abs().max())
        assert float(o[1, 7:40].abs().max()) == 0.0
    assert float((x.grad.float() - xr.grad).abs().max()) < 3e-2 * float(xr.grad.abs().max())
    for w, w2, b, b2 in zip(ws, wr, bs_, br):
        assert float((w.grad - w2.grad).abs().max()) < 3e-2 * float(w2.grad.abs().max())
        assert float((b.grad - b2.grad).abs().max()) < 3e-2 * float(b2.grad.abs().max())
    # the single projection with the mask in its epilogue
    y = Lin256Function.apply(x.detach(), pack_linear256(ws[:1], bs_[:1]), mask, False, ws[0], bs_[0])
    assert float((y.float() - want[0].detach()).abs().max()) < 2e-2 * float(want[0].abs().max())


def _decoder_pair(layers=2):
    from richsem_amd.modules import MLP, DeformableTransformerDecoderLayer, TransformerDecoder
    torch.manual_seed(11)
    layer = DeformableTransformerDecoderLayer(256, 512, dropout=0.0, activation="relu", n_levels=4, n_heads=8, n_points=4)
    dec = TransformerDecoder(layer, layers, torch.nn.LayerNorm(256), d_model=256)
    dec.bbox_embed = torch.nn.ModuleList([MLP(256, 256, 4, 3) for _ in range(layers)])
    dec = dec.cuda()
    with torch.no_grad():
        for l in dec.layers:
            l.cross_attn.sampling_offsets.weight.normal_(0, 0.01)
            l.cross_attn.attention_weights.weight.normal_(0, 0.05)
    return dec


# mean |bf16 - fp32| / mean |fp32| of the bf16 layer paths against their own fp32 path: at most 2 x what MI355X measures
# (profiles/r04_bf16_bounds.txt: decoder hs 4.7e-3, refs 1.7e-4, input gradients 7.6e-2 / 5.4e-2, worst parameter gradient 0.139 --
# sampling_offsets of the second layer --; encoder out 3.6e-3, grad_src 1.4e-2, grad_pos 3.6e-2, worst parameter gradient 0.105)
HS_TOL, REFS_TOL, GRAD_IN_TOL, GRAD_PARAM_TOL = 1e-2, 5e-4, 0.12, 0.25
ENC_OUT_TOL, ENC_GRAD_SRC_TOL, ENC_GRAD_POS_TOL, ENC_PARAM_TOL = 8e-3, 3e-2, 7.5e-2, 0.21


def test_bf16_decoder_on_the_library_kernels_is_close_to_its_fp32_path():
    """the bf16 decoder (lin256 projections, attention kernels, stacked value projection, fused add + LayerNorm, small-token FFN) against the
    same module's fp32 path -- the op sequence the reference-generated fixtures above pin -- on a shrunk decoder-shaped call"""
    from richsem_amd import workload as W
    call = W.shrunk(W.call_Dd(2), 4)
    shapes, lsi = W.level_tensors(call, "cuda")
    dec = _decoder_pair()
    nq, bs, S = 150, call.N, call.S
    g = torch.Generator(device="cuda").manual_seed(5)
    tgt = torch.randn(nq, bs, 256, device="cuda", generator=g)
    mem = torch.randn(S, bs, 256, device="cuda", generator=g)
    refu = torch.randn(nq, bs, 4, device="cuda", generator=g)
    vr = torch.rand(bs, 4, 2, device="cuda", generator=g) * 0.2 + 0.8
    mmask = torch.zeros(bs, S, dtype=torch.bool, device="cuda")
    mmask[1, -9:] = True
    amask = torch.zeros(nq, nq, dtype=torch.bool, device="cuda")
    amask[40:, :40] = True
    amask[:20, 20:40] = True
    amask[20:40, :20] = True
    res = {}
    for dt in (torch.float32, torch.bfloat16):
        a, m = tgt.detach().clone().to(dt).requires_grad_(True), mem.detach().clone().to(dt).requires_grad_(True)
        r = refu.detach().clone().requires_grad_(True)
        hs, refs = dec(tgt=a, memory=m, tgt_mask=amask, memory_key_padding_mask=mmask, refpoints_unsigmoid=r, level_start_index=lsi,
                       spatial_shapes=shapes, valid_ratios=vr)
        hs, refs = torch.stack(hs).float(), torch.stack(refs).float()
        (hs.square().mean() + refs.square().mean()).backward()
        res[dt] = (hs.detach(), refs.detach(), a.grad.float(), m.grad.float(), {k: p.grad.clone() for k, p in dec.named_parameters() if p.grad is not None})
        dec.zero_grad()
    (h32, r32, ga32, gm32, gp32), (h16, r16, ga16, gm16, gp16) = res[torch.float32], res[torch.bfloat16]
    assert _ratio(h16, h32, "decoder hs") < HS_TOL and _ratio(r16, r32, "decoder refs") < REFS_TOL
    assert _ratio(ga16, ga32, "decoder grad_tgt") < GRAD_IN_TOL and _ratio(gm16, gm32, "decoder grad_memory") < GRAD_IN_TOL
    worst = max((_ratio(gp16[k].float(), gp32[k]), k) for k in gp32 if gp32[k].abs().mean() > 0)
    _ratio(gp16[worst[1]].float(), gp32[worst[1]], "decoder worst parameter gradient " + worst[1])
    assert worst[0] < GRAD_PARAM_TOL, worst


def test_grouped_layer_weight_gradients_equal_the_separate_launches():
    """functions/linear.py: WgradGroup / WgradBoundary -- a decoder layer's seven weight (+ bias) gradients deferred to ONE launch behind
    the layer's backward -- against one launch per linear layer: equal outputs and input gradients (nothing else changes), parameter
    gradients equal up to the order of the pixel-chunk sums"""
    from richsem_amd import workload as W
    from richsem_amd.functions.linear import WgradGroup
    call = W.shrunk(W.call_Dd(2), 4)
    shapes, lsi = W.level_tensors(call, "cuda")
    dec = _decoder_pair()
    nq, bs, S = 640, call.N, call.S                     # 1280 tokens: above the weight-gradient kernel's threshold
    g = torch.Generator(device="cuda").manual_seed(6)
    tgt = torch.randn(nq, bs, 256, device="cuda", generator=g).to(torch.bfloat16)
    mem = torch.randn(S, bs, 256, device="cuda", generator=g).to(torch.bfloat16)
    refu = torch.randn(nq, bs, 4, device="cuda", generator=g)
    vr = torch.rand(bs, 4, 2, device="cuda", generator=g) * 0.2 + 0.8
    amask = torch.zeros(nq, nq, dtype=torch.bool, device="cuda")
    amask[40:, :40] = True
    res = {}
    for grouped in (True, False):
        WgradGroup.enabled = grouped
        try:
            a, m, r = tgt.clone().requires_grad_(True), mem.clone().requires_grad_(True), refu.clone().requires_grad_(True)
            hs, refs = dec(tgt=a, memory=m, tgt_mask=amask, memory_key_padding_mask=None, refpoints_unsigmoid=r, level_start_index=lsi,
                           spatial_shapes=shapes, valid_ratios=vr)
            hs, refs = torch.stack(hs).float(), torch.stack(refs).float()
            (hs.square().mean() + refs.square().mean()).backward()
            res[grouped] = (hs.detach(), a.grad.float(), m.grad.float(), {k: p.grad.clone() for k, p in dec.named_parameters() if p.grad is not None})
        finally:
            WgradGroup.enabled = True
            dec.zero_grad()
    (h1, ga1, gm1, gp1), (h0, ga0, gm0, gp0) = res[True], res[False]
    assert torch.equal(h1, h0) and torch.equal(ga1, ga0) and torch.equal(gm1, gm0)
    assert sorted(gp1) == sorted(gp0)
    for k in gp0:
        assert torch.isfinite(gp1[k]).all(), k
        assert float((gp1[k].float() - gp0[k].float()).abs().max()) <= 1e-4 * float(gp0[k].float().abs().max()) + 1e-12, k


def test_lin256_on_raw_parameters_inside_an_active_group_is_not_deferred():
    """round-4 advice: `Lin256Function` called on RAW module parameters while some layer's WgradGroup is active must not hand autograd an
    unwritten gradient (AccumulateGrad could run before the group's flush): only the group's own boundary aliases are deferred.  Same
    weight gradient as without any group, and nothing left pending."""
    from richsem_amd.functions.linear import Lin256Function, WgradGroup, pack_linear256
    torch.manual_seed(4)
    lin = torch.nn.Linear(256, 256).cuda()
    x = torch.randn(2048, 256, device="cuda").to(torch.bfloat16)
    pk = pack_linear256([lin.weight], [lin.bias])
    Lin256Function.apply(x, pk, None, False, lin.weight, lin.bias).float().square().mean().backward()
    want_w, want_b = lin.weight.grad.clone(), lin.bias.grad.clone()
    lin.zero_grad()
    group = WgradGroup()
    with group:
        y = Lin256Function.apply(x, pk, None, False, lin.weight, lin.bias)
    y.float().square().mean().backward()
    assert not group.pending
    assert torch.equal(lin.weight.grad, want_w) and torch.equal(lin.bias.grad, want_b)


def test_bf16_encoder_layer_close_to_fp32_path():
    """the bf16 ENCODER layer at the shipped width (d_model 256, d_ffn 2048, 8 heads) with the DEFAULT thresholds -- 16800 tokens: lin256
    projections with the padding mask in the epilogue, the operator's bf16 entry points (routed backward), add + LayerNorm kernel, the
    fused feed-forward kernel and its backward -- against the same layer's fp32 op sequence, which the reference-class fixtures pin
    (round-3 verdict: the largest row of the composed step had no numeric layer-level test)"""
    from richsem_amd import workload as W
    from richsem_amd.modules import DeformableTransformerEncoderLayer, get_reference_points
    from richsem_amd.functions.ffn import FUSED_FFN_MIN_TOKENS
    call = W.shrunk(W.call_E(3), 2)                       # 3 images of the half-size pyramid: 3 x 5600 tokens
    assert call.N * call.S >= FUSED_FFN_MIN_TOKENS
    shapes, lsi = W.level_tensors(call, "cuda")
    torch.manual_seed(21)
    layer = DeformableTransformerEncoderLayer(256, 2048, dropout=0.0, activation="relu", n_levels=4, n_heads=8, n_points=4).cuda()
    with torch.no_grad():
        layer.self_attn.sampling_offsets.weight.normal_(0, 0.01)
        layer.self_attn.attention_weights.weight.normal_(0, 0.05)
    g = torch.Generator(device="cuda").manual_seed(6)
    src = torch.randn(call.N, call.S, 256, device="cuda", generator=g)
    pos = torch.randn(call.N, call.S, 256, device="cuda", generator=g)
    vr = torch.rand(call.N, 4, 2, device="cuda", generator=g) * 0.2 + 0.8
    ref = get_reference_points(shapes.tolist(), vr, "cuda")
    mask = torch.zeros(call.N, call.S, dtype=torch.bool, device="cuda")
    mask[1, 80:84] = True
    mask[2, -5:] = True
    go = torch.randn(call.N, call.S, 256, device="cuda", generator=g)
    res = {}
    for dt in (torch.float32, torch.bfloat16):
        a, p = src.clone().to(dt).requires_grad_(True), pos.clone().to(dt).requires_grad_(True)
        out = layer(a, p, ref, shapes, lsi, mask)
        assert out.dtype == dt
        out.backward(go.to(dt))
        res[dt] = (out.detach().float(), a.grad.float(), p.grad.float(), {k: q.grad.clone() for k, q in layer.named_parameters()})
        layer.zero_grad()
    (o32, a32, p32, g32), (o16, a16, p16, g16) = res[torch.float32], res[torch.bfloat16]
    assert _ratio(o16, o32, "encoder out") < ENC_OUT_TOL
    assert _ratio(a16, a32, "encoder grad_src") < ENC_GRAD_SRC_TOL and _ratio(p16, p32, "encoder grad_pos") < ENC_GRAD_POS_TOL
    worst = max((_ratio(g16[k].float(), g32[k]), k) for k in g32)
    _ratio(g16[worst[1]].float(), g32[worst[1]], "encoder worst parameter gradient " + worst[1])
    assert worst[0] < ENC_PARAM_TOL, worst


def test_bf16_layer_follows_an_optimizer_step():
    """the kept bf16 / packed forms of the parameters (VersionCache) must notice in-place updates: forward, SGD step, forward again --
    for the attention module's cache and for the decoder layer's"""
    from richsem_amd import workload as W
    from richsem_amd.modules import MSDeformAttn
    call = W.shrunk(W.call_Dd(2), 4)
    shapes, lsi = W.level_tensors(call, "cuda")
    torch.manual_seed(4)
    mod = MSDeformAttn(256, 4, 8, 4).cuda()
    ref_mod = MSDeformAttn(256, 4, 8, 4).cuda()
    q = torch.randn(call.N, 50, 256, device="cuda").to(torch.bfloat16)
    src = torch.randn(call.N, call.S, 256, device="cuda").to(torch.bfloat16)
    rp = torch.rand(call.N, 50, 4, 4, device="cuda") * 0.4 + 0.2
    opt = torch.optim.SGD(mod.parameters(), lr=0.5)
    for step in range(2):
        out = mod(q, rp, src, shapes, lsi, None)
        ref_mod.load_state_dict(mod.state_dict())          # a fresh module: nothing cached
        want = ref_mod(q, rp, src, shapes, lsi, None)
        assert torch.equal(out, want), step
        opt.zero_grad()
        out.float().square().mean().backward()
        before = mod.value_proj.weight.detach().clone()
        opt.step()
        assert not torch.equal(before, mod.value_proj.weight.detach())
    # a write through .data is invisible to the version counter: invalidate_bf16_cache() is the documented way
    mod.value_proj.weight.data.mul_(0.5)
    mod.invalidate_bf16_cache()
    ref_mod.load_state_dict(mod.state_dict())
    assert torch.equal(mod(q, rp, src, shapes, lsi, None), ref_mod(q, rp, src, shapes, lsi, None))


@pytest.mark.parametrize("dims", [2, 4])
def test_sine_embed_kernel_matches_the_mirror(dims):
    """msda_sine_embed_bf16 (one launch) against gen_sineembed_for_position (the reference's op sequence, fp32) to one bf16 rounding,
    on contiguous boxes and on the level-0 slice of the (bs, nq, L, dims) boxes the decoder hands it"""
    from richsem_amd.modules.decoder import gen_sineembed_for_position, sine_embed_bf16
    g = torch.Generator(device="cuda").manual_seed(3)
    full = torch.rand(2, 137, 4, dims, device="cuda", generator=g)
    for boxes in (full[:, :, 0, :], full[:, :, 0, :].contiguous()):
        got = sine_embed_bf16(boxes, 128).float()
        ref = gen_sineembed_for_position(boxes.contiguous(), 128)
        assert got.shape == ref.shape
        assert (got - ref).abs().max() <= 2 ** -8 + 1e-5


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_refine_boxes_kernel_matches_the_op_sequence(dt):
    """msda_box_refine_* against (delta + inverse_sigmoid(ref)).sigmoid() and its autograd gradient, incl. boxes at / beyond 0 and 1"""
    from richsem_amd.modules.decoder import inverse_sigmoid, refine_boxes
    g = torch.Generator(device="cuda").manual_seed(9)
    ref = torch.rand(2, 300, 4, device="cuda", generator=g)
    ref[0, :5] = torch.tensor([0.0, 1.0, 1e-4, 1 - 1e-4], device="cuda")
    ref[1, 0] = torch.tensor([-0.1, 1.2, 0.5, 0.5], device="cuda")
    delta = (torch.randn(2, 300, 4, device="cuda", generator=g)).to(dt).requires_grad_(True)
    gy = torch.randn(2, 300, 4, device="cuda", generator=g)
    y = refine_boxes(delta, ref)
    y.backward(gy)
    d2 = delta.detach().clone().requires_grad_(True)
    y2 = (d2.float() + inverse_sigmoid(ref)).sigmoid()
    y2.backward(gy)
    assert y.dtype == torch.float32 and (y - y2).abs().max() < 2e-6
    tol = 2e-6 if dt == torch.float32 else 2 ** -8
    assert (delta.grad.float() - d2.grad.float()).abs().max() <= tol * max(1.0, float(d2.grad.float().abs().max()))
    # a reference that carries a gradient (the heads' boxes of decoder layers 1..5): inverse_sigmoid's clamps differentiated as torch does,
    # incl. the values on and beyond the bounds
    r1, r2 = ref.clone().requires_grad_(True), ref.clone().requires_grad_(True)
    d1, d3 = delta.detach().clone().requires_grad_(True), delta.detach().clone().requires_grad_(True)
    refine_boxes(d1, r1).backward(gy)
    (d3.float() + inverse_sigmoid(r2)).sigmoid().backward(gy)
    assert torch.isfinite(r1.grad).all()
    scale = float(r2.grad.abs().max())
    assert float((r1.grad - r2.grad).abs().max()) <= 2e-5 * scale, float((r1.grad - r2.grad).abs().max()) / scale
    assert (d1.grad.float() - d3.grad.float()).abs().max() <= tol * max(1.0, float(d3.grad.float().abs().max()))


@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_bf16_layers_on_small_inputs_and_with_dropout(dropout):
    """bf16 activations with fp32 parameters below the fused kernels' token thresholds (DEFAULT thresholds) and with dropout active in
    train mode: the op-by-op paths must cast the parameters (they used to raise a dtype mismatch)"""
    from richsem_amd.modules import DeformableTransformerDecoderLayer, DeformableTransformerEncoderLayer, get_reference_points
    torch.manual_seed(0)
    shapes_l = [(12, 16), (6, 8), (3, 4), (2, 2)]
    shapes = torch.tensor(shapes_l, dtype=torch.int64, device="cuda")
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    S, N = int(shapes.prod(1).sum()), 2
    enc = DeformableTransformerEncoderLayer(256, 512, dropout=dropout, n_levels=4, n_heads=8, n_points=4).cuda().train()
    dec = DeformableTransformerDecoderLayer(256, 512, dropout=dropout, n_levels=4, n_heads=8, n_points=4).cuda().train()
    src = torch.randn(N, S, 256, device="cuda").to(torch.bfloat16).requires_grad_(True)
    pos = torch.randn(N, S, 256, device="cuda").to(torch.bfloat16)
    vr = torch.ones(N, 4, 2, device="cuda")
    mem = enc(src, pos, get_reference_points(shapes_l, vr, "cuda"), shapes, lsi, None)
    assert mem.dtype == torch.bfloat16 and torch.isfinite(mem.float()).all()
    nq = 20
    tgt = torch.randn(nq, N, 256, device="cuda").to(torch.bfloat16).requires_grad_(True)
    out = dec(tgt=tgt, tgt_query_pos=torch.randn(nq, N, 256, device="cuda").to(torch.bfloat16),
              tgt_reference_points=torch.rand(nq, N, 4, 4, device="cuda") * 0.5 + 0.2, memory=mem.transpose(0, 1),
              memory_level_start_index=lsi, memory_spatial_shapes=shapes)
    assert out.dtype == torch.bfloat16 and torch.isfinite(out.float()).all()
    out.float().square().mean().backward()
    assert src.grad is not None and tgt.grad is not None and torch.isfinite(src.grad.float()).all()


@pytest.mark.parametrize("T", [2184, 44646, 37])
def test_narrow_linear_function_matches_autograd(T):
    """Lin256NarrowFunction (256 -> 4: forward on lin256 with a padded weight, backward on the two streaming kernels) against F.linear in
    fp32 autograd on the same bf16 inputs"""
    from richsem_amd.functions.linear import Lin256NarrowFunction, pack_linear256_padded
    g = torch.Generator(device="cuda").manual_seed(T)
    x = torch.randn(T, 256, device="cuda", generator=g).to(torch.bfloat16).requires_grad_(True)
    w = (torch.randn(4, 256, device="cuda", generator=g) / 16).requires_grad_(True)
    b = torch.randn(4, device="cuda", generator=g).requires_grad_(True)
    dy = torch.randn(T, 4, device="cuda", generator=g).to(torch.bfloat16)
    y = Lin256NarrowFunction.apply(x, pack_linear256_padded(w, b), w, b)
    y.backward(dy)
    xr, br = x.detach().float().requires_grad_(True), b.detach().clone().requires_grad_(True)
    wr = w.detach().to(torch.bfloat16).float().requires_grad_(True)      # (the forward multiplies with the bf16-rounded weight)
    yr = torch.nn.functional.linear(xr, wr, br)
    yr.backward(dy.float())
    assert (y.float() - yr).abs().max() <= 2 ** -7 * yr.abs().max()
    assert (x.grad.float() - xr.grad).abs().max() <= 2 ** -7 * xr.grad.abs().max()
    assert (w.grad - wr.grad).abs().max() <= 1e-3 * wr.grad.abs().max()
    assert (b.grad - br.grad).abs().max() <= 1e-3 * br.grad.abs().max()
