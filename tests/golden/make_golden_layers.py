#!/usr/bin/env python3
"""Golden vectors for SURVEY.md section 8a rows a9 / a12, generated from the REFERENCE's own classes and functions.

Run in the build container only (it reads /root/reference; the fixtures it writes are committed, the GPU box never sees the
reference):

    python tests/golden/make_golden_layers.py

What is executed is the reference's code:
  * ``models/richsem/deformable_transformer.py`` is loaded by file path into a synthetic package (``_refrs``) -- the package's own
    ``__init__`` drags in torchvision / timm / detectron2, which are absent here.  Its relative imports are served by
      - ``_refrs.utils``              = the reference's ``models/richsem/utils.py``, loaded by path (pure torch);
      - ``_refrs.ops.modules``        = the reference's ``ops/modules/ms_deform_attn.py``, loaded by path, with
        ``MSDeformAttnFunction.apply`` bound to the reference's pure-PyTorch core ``ms_deform_attn_core_pytorch``
        (as tests/golden/make_golden.py does for the module fixtures);
      - ``util.misc.inverse_sigmoid`` = that function's own source, cut out of ``util/misc.py`` (util/misc.py:605-609) with
        ``ast`` and executed -- the file as a whole imports torchvision.
  * ``models/richsem/dn_components.py`` the same way (``prepare_for_cdn``); it calls ``.cuda()`` / ``.to('cuda')``
    unconditionally (dn_components.py:28,68-69,131-137,158), so while it runs ``torch.Tensor.cuda`` and ``.to('cuda')`` are mapped to
    the CPU tensor itself (a shim of torch's device placement, not of the reference).
Everything is float64 (layers) / int64 + bool (denoising set-up), seeded.
"""
import ast
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _load(name, path, package=None):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    if package:
        m.__package__ = package
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def _functions_of(path, names):
    """the named top-level functions of a reference source file, executed on their own (the file's imports are not)"""
    tree = ast.parse(open(path).read())
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(body) == len(names), (path, names)
    ns = {"torch": torch}
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return {n: ns[n] for n in names}


def reference_modules():
    # the pure-PyTorch core the reference ships (ops/functions/ms_deform_attn_func.py:41-61)
    sys.modules.setdefault("MultiScaleDeformableAttention", types.ModuleType("MultiScaleDeformableAttention"))
    core = _load("_ref_func", f"{REF}/models/richsem/ops/functions/ms_deform_attn_func.py").ms_deform_attn_core_pytorch

    pkg = types.ModuleType("_refrs"); pkg.__path__ = []
    ops = types.ModuleType("_refrs.ops"); ops.__path__ = []
    fn_mod = types.ModuleType("_refrs.ops.functions")

    class MSDeformAttnFunction:   # same call signature as the reference's autograd Function (ms_deform_attn_func.py:23)
        @staticmethod
        def apply(value, shapes, lsi, loc, aw, im2col_step):
            return core(value, shapes, loc, aw)

    fn_mod.MSDeformAttnFunction = MSDeformAttnFunction
    mods = types.ModuleType("_refrs.ops.modules"); mods.__path__ = []
    sys.modules.update({"_refrs": pkg, "_refrs.ops": ops, "_refrs.ops.functions": fn_mod, "_refrs.ops.modules": mods})
    msda = _load("_refrs.ops.modules.ms_deform_attn", f"{REF}/models/richsem/ops/modules/ms_deform_attn.py", "_refrs.ops.modules")
    mods.MSDeformAttn = msda.MSDeformAttn
    _load("_refrs.utils", f"{REF}/models/richsem/utils.py", "_refrs")

    util = types.ModuleType("util"); util.__path__ = []
    misc = types.ModuleType("util.misc")
    misc.inverse_sigmoid = _functions_of(f"{REF}/util/misc.py", ["inverse_sigmoid"])["inverse_sigmoid"]
    for unused in ("NestedTensor", "nested_tensor_from_tensor_list", "accuracy", "get_world_size", "interpolate",
                   "is_dist_avail_and_initialized"):     # names dn_components.py imports and prepare_for_cdn never touches
        setattr(misc, unused, None)
    box_ops = types.ModuleType("util.box_ops")
    box_ops.box_cxcywh_to_xyxy = _functions_of(f"{REF}/util/box_ops.py", ["box_cxcywh_to_xyxy"])["box_cxcywh_to_xyxy"]
    box_ops.box_iou = None                                # (only under check_pos_dn=True; needs torchvision)
    util.misc, util.box_ops = misc, box_ops
    sys.modules.update({"util": util, "util.misc": misc, "util.box_ops": box_ops})
    dt = _load("_refrs.deformable_transformer", f"{REF}/models/richsem/deformable_transformer.py", "_refrs")
    dn = _load("_refrs.dn_components", f"{REF}/models/richsem/dn_components.py", "_refrs")
    return dt, dn, sys.modules["_refrs.utils"]


SHAPES = [(9, 14), (5, 7), (3, 4), (2, 2)]


def pyramid():
    shapes = torch.as_tensor(SHAPES, dtype=torch.long)
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    return shapes, lsi, int(shapes.prod(1).sum())


def perturb(mod, gen):
    """move every parameter away from its (partly all-zero) initial value so that each one matters"""
    with torch.no_grad():
        for p in mod.parameters():
            p.add_(torch.randn(p.shape, generator=gen, dtype=p.dtype) * 0.05)


def save(name, arrays, mod):
    sd = {"param." + k: v.detach().numpy() for k, v in mod.state_dict().items()}
    grads = {"param." + k + ".grad": p.grad.numpy() for k, p in mod.named_parameters() if p.grad is not None}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **{k: (v.detach().numpy() if torch.is_tensor(v) else v) for k, v in arrays.items()},
                        **sd, **grads)
    print(name, {k: tuple(v.shape) for k, v in arrays.items() if torch.is_tensor(v)})


def encoder_layer(dt):
    shapes, lsi, S = pyramid()
    N, C, H, L, P, F = 2, 64, 2, 4, 4, 96
    gen = torch.Generator().manual_seed(301)
    torch.manual_seed(300)
    layer = dt.DeformableTransformerEncoderLayer(d_model=C, d_ffn=F, dropout=0.0, activation="relu", n_levels=L, n_heads=H, n_points=P).double()
    perturb(layer, gen)
    src = torch.randn(N, S, C, generator=gen, dtype=torch.float64, requires_grad=True)
    pos = torch.randn(N, S, C, generator=gen, dtype=torch.float64, requires_grad=True)
    valid_ratios = torch.rand(N, L, 2, generator=gen, dtype=torch.float64) * 0.3 + 0.7
    ref = dt.TransformerEncoder.get_reference_points(shapes, valid_ratios, device="cpu").double()      # :512-525
    mask = torch.zeros(N, S, dtype=torch.bool)
    mask[1, lsi[0] + 9:lsi[0] + 14] = True
    mask[1, -1] = True
    out = layer(src, pos, ref, shapes, lsi, mask)
    go = torch.randn(out.shape, generator=gen, dtype=torch.float64)
    out.backward(go)
    save("layer_encoder_f64", dict(src=src, pos=pos, valid_ratios=valid_ratios, reference_points=ref, shapes=shapes, lsi=lsi, mask=mask,
                                   out=out, grad_out=go, grad_src=src.grad, grad_pos=pos.grad), layer)


def decoder_layer(dt):
    shapes, lsi, S = pyramid()
    N, C, H, L, P, F, nq = 2, 64, 2, 4, 4, 96, 23
    gen = torch.Generator().manual_seed(311)
    torch.manual_seed(310)
    layer = dt.DeformableTransformerDecoderLayer(d_model=C, d_ffn=F, dropout=0.0, activation="relu", n_levels=L, n_heads=H, n_points=P,
                                                 decoder_sa_type="sa", module_seq=["sa", "ca", "ffn"]).double()     # baseline_4scale.py
    perturb(layer, gen)
    tgt = torch.randn(nq, N, C, generator=gen, dtype=torch.float64, requires_grad=True)
    qpos = torch.randn(nq, N, C, generator=gen, dtype=torch.float64, requires_grad=True)
    memory = torch.randn(S, N, C, generator=gen, dtype=torch.float64, requires_grad=True)
    cxcy = torch.rand(nq, N, 1, 2, generator=gen, dtype=torch.float64) * 0.8 + 0.1
    wh = torch.rand(nq, N, 1, 2, generator=gen, dtype=torch.float64) * 0.4 + 0.05
    ref = torch.cat((cxcy, wh), -1).expand(nq, N, L, 4).contiguous()
    mmask = torch.zeros(N, S, dtype=torch.bool)
    mmask[0, lsi[1]:lsi[1] + 6] = True
    amask = torch.zeros(nq, nq, dtype=torch.bool)       # a denoising-style self-attention mask: two groups + matching part
    amask[8:, :8] = True
    amask[:4, 4:8] = True
    amask[4:8, :4] = True
    out = layer(tgt=tgt, tgt_query_pos=qpos, tgt_reference_points=ref, memory=memory, memory_key_padding_mask=mmask,
                memory_level_start_index=lsi, memory_spatial_shapes=shapes, self_attn_mask=amask)
    go = torch.randn(out.shape, generator=gen, dtype=torch.float64)
    out.backward(go)
    save("layer_decoder_f64", dict(tgt=tgt, query_pos=qpos, memory=memory, reference_points=ref, shapes=shapes, lsi=lsi, memory_mask=mmask,
                                   attn_mask=amask, out=out, grad_out=go, grad_tgt=tgt.grad, grad_query_pos=qpos.grad,
                                   grad_memory=memory.grad), layer)


def decoder_stack(dt, ut):
    """TransformerDecoder.forward (:693-823) in the shipped configuration: deformable decoder, 4-d reference boxes, iterative box
    refinement through ``bbox_embed`` (one MLP per layer, as richsem.py attaches them), look-forward-twice reference lists."""
    shapes, lsi, S = pyramid()
    N, C, H, L, P, F, nq, nl = 2, 64, 2, 4, 4, 96, 19, 3
    gen = torch.Generator().manual_seed(321)
    torch.manual_seed(320)
    layer = dt.DeformableTransformerDecoderLayer(d_model=C, d_ffn=F, dropout=0.0, activation="relu", n_levels=L, n_heads=H, n_points=P,
                                                 decoder_sa_type="sa", module_seq=["sa", "ca", "ffn"])
    dec = dt.TransformerDecoder(layer, nl, torch.nn.LayerNorm(C), return_intermediate=True, d_model=C, query_dim=4,
                                num_feature_levels=L, deformable_decoder=True, rm_dec_query_scale=True)
    dec.bbox_embed = torch.nn.ModuleList([ut.MLP(C, C, 4, 3) for _ in range(nl)])       # richsem.py: box heads handed to the decoder
    dec = dec.double()
    perturb(dec, gen)
    tgt = torch.randn(nq, N, C, generator=gen, dtype=torch.float64, requires_grad=True)
    memory = torch.randn(S, N, C, generator=gen, dtype=torch.float64, requires_grad=True)
    refu = torch.randn(nq, N, 4, generator=gen, dtype=torch.float64, requires_grad=True)
    valid_ratios = torch.rand(N, L, 2, generator=gen, dtype=torch.float64) * 0.3 + 0.7
    mmask = torch.zeros(N, S, dtype=torch.bool)
    mmask[1, lsi[2]:lsi[2] + 3] = True
    amask = torch.zeros(nq, nq, dtype=torch.bool)
    amask[6:, :6] = True
    hs, refs = dec(tgt=tgt, memory=memory, tgt_mask=amask, memory_key_padding_mask=mmask, pos=None, refpoints_unsigmoid=refu,
                   level_start_index=lsi, spatial_shapes=shapes, valid_ratios=valid_ratios)
    hs, refs = torch.stack(hs), torch.stack(refs)
    g_hs = torch.randn(hs.shape, generator=gen, dtype=torch.float64)
    g_refs = torch.randn(refs.shape, generator=gen, dtype=torch.float64)
    ((hs * g_hs).sum() + (refs * g_refs).sum()).backward()
    save("decoder_stack_f64", dict(tgt=tgt, memory=memory, refpoints_unsigmoid=refu, valid_ratios=valid_ratios, shapes=shapes, lsi=lsi,
                                   memory_mask=mmask, attn_mask=amask, hs=hs, refs=refs, grad_hs=g_hs, grad_refs=g_refs,
                                   grad_tgt=tgt.grad, grad_memory=memory.grad, grad_refpoints=refu.grad), dec)


# ---- d_model = 256, 8 heads (round 4): the shapes the bf16 kernels are built for (K = 256 projections, 8 x 32-channel attention heads,
#      the fused feed-forward block), so that they meet reference-class outputs directly.  Parameters come from tests/layer_params.py (a
#      pure function of seed and key, rebuilt by the test); stored: inputs, outputs, input gradients, parameter gradients as float16
#      mantissas + scale.
def _np(t):
    """float64 activations are stored as float32 (the inputs are float32 values to begin with: `_rand32`): these fixtures are compared at
    bf16 tolerance"""
    if not torch.is_tensor(t):
        return t
    t = t.detach()
    return (t.float() if t.dtype == torch.float64 else t).numpy()


def _rand32(fn, *shape, gen, requires_grad=False):
    """float32 values, held in float64: what the fixture stores is exactly what the reference ran on"""
    return fn(*shape, generator=gen, dtype=torch.float32).double().requires_grad_(requires_grad)


def layers_256(dt, ut):
    sys.path.insert(0, os.path.dirname(OUT))
    import layer_params as LP
    shapes = torch.as_tensor(LP.SHAPES, dtype=torch.long)
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    N, C, H, L, P, F = 2, LP.D_MODEL, LP.HEADS, LP.LEVELS, LP.POINTS, LP.D_FFN
    # -- encoder layer
    gen = torch.Generator().manual_seed(401)
    layer = LP.fill(dt.DeformableTransformerEncoderLayer(d_model=C, d_ffn=F, dropout=0.0, activation="relu", n_levels=L, n_heads=H,
                                                         n_points=P).double(), 400)
    src = _rand32(torch.randn, N, S, C, gen=gen, requires_grad=True)
    pos = _rand32(torch.randn, N, S, C, gen=gen, requires_grad=True)
    valid_ratios = _rand32(torch.rand, N, L, 2, gen=gen) * 0.3 + 0.7
    ref = dt.TransformerEncoder.get_reference_points(shapes, valid_ratios, device="cpu").double()
    mask = torch.zeros(N, S, dtype=torch.bool)
    mask[1, lsi[0] + 8:lsi[0] + 12] = True
    mask[1, -2:] = True
    out = layer(src, pos, ref, shapes, lsi, mask)
    go = _rand32(torch.randn, out.shape, gen=gen)
    out.backward(go)
    arrays = dict(src=src, pos=pos, valid_ratios=valid_ratios, reference_points=ref, shapes=shapes, lsi=lsi, mask=mask, out=out, grad_out=go,
                  grad_src=src.grad, grad_pos=pos.grad)
    np.savez_compressed(os.path.join(OUT, "layer256_encoder_f64.npz"), **{k: _np(v) for k, v in arrays.items()}, **LP.pack_grads(layer))
    print("layer256_encoder_f64", tuple(out.shape))
    # -- decoder layer
    nq = 40
    gen = torch.Generator().manual_seed(411)
    layer = LP.fill(dt.DeformableTransformerDecoderLayer(d_model=C, d_ffn=F, dropout=0.0, activation="relu", n_levels=L, n_heads=H, n_points=P,
                                                         decoder_sa_type="sa", module_seq=["sa", "ca", "ffn"]).double(), 410)
    tgt = _rand32(torch.randn, nq, N, C, gen=gen, requires_grad=True)
    qpos = _rand32(torch.randn, nq, N, C, gen=gen, requires_grad=True)
    memory = _rand32(torch.randn, S, N, C, gen=gen, requires_grad=True)
    cxcy = _rand32(torch.rand, nq, N, 1, 2, gen=gen) * 0.8 + 0.1
    wh = _rand32(torch.rand, nq, N, 1, 2, gen=gen) * 0.4 + 0.05
    ref = torch.cat((cxcy, wh), -1).expand(nq, N, L, 4).contiguous()
    mmask = torch.zeros(N, S, dtype=torch.bool)
    mmask[0, lsi[1]:lsi[1] + 6] = True
    amask = torch.zeros(nq, nq, dtype=torch.bool)       # denoising-style: two groups of 6 that do not see each other, the matching part sees neither
    amask[12:, :12] = True
    amask[:6, 6:12] = True
    amask[6:12, :6] = True
    out = layer(tgt=tgt, tgt_query_pos=qpos, tgt_reference_points=ref, memory=memory, memory_key_padding_mask=mmask,
                memory_level_start_index=lsi, memory_spatial_shapes=shapes, self_attn_mask=amask)
    go = _rand32(torch.randn, out.shape, gen=gen)
    out.backward(go)
    arrays = dict(tgt=tgt, query_pos=qpos, memory=memory, reference_points=ref, shapes=shapes, lsi=lsi, memory_mask=mmask, attn_mask=amask,
                  out=out, grad_out=go, grad_tgt=tgt.grad, grad_query_pos=qpos.grad, grad_memory=memory.grad)
    np.savez_compressed(os.path.join(OUT, "layer256_decoder_f64.npz"), **{k: _np(v) for k, v in arrays.items()}, **LP.pack_grads(layer))
    print("layer256_decoder_f64", tuple(out.shape))
    # -- two-layer decoder with box refinement
    nl = 2
    gen = torch.Generator().manual_seed(421)
    layer = dt.DeformableTransformerDecoderLayer(d_model=C, d_ffn=F, dropout=0.0, activation="relu", n_levels=L, n_heads=H, n_points=P,
                                                 decoder_sa_type="sa", module_seq=["sa", "ca", "ffn"])
    dec = dt.TransformerDecoder(layer, nl, torch.nn.LayerNorm(C), return_intermediate=True, d_model=C, query_dim=4,
                                num_feature_levels=L, deformable_decoder=True, rm_dec_query_scale=True)
    dec.bbox_embed = torch.nn.ModuleList([ut.MLP(C, C, 4, 3) for _ in range(nl)])
    dec = LP.fill(dec.double(), 420)
    tgt = _rand32(torch.randn, nq, N, C, gen=gen, requires_grad=True)
    memory = _rand32(torch.randn, S, N, C, gen=gen, requires_grad=True)
    refu = _rand32(torch.randn, nq, N, 4, gen=gen, requires_grad=True)
    valid_ratios = _rand32(torch.rand, N, L, 2, gen=gen) * 0.3 + 0.7
    mmask = torch.zeros(N, S, dtype=torch.bool)
    mmask[1, lsi[2]:lsi[2] + 2] = True
    hs, refs = dec(tgt=tgt, memory=memory, tgt_mask=amask, memory_key_padding_mask=mmask, pos=None, refpoints_unsigmoid=refu,
                   level_start_index=lsi, spatial_shapes=shapes, valid_ratios=valid_ratios)
    hs, refs = torch.stack(hs), torch.stack(refs)
    g_hs = _rand32(torch.randn, hs.shape, gen=gen)
    g_refs = _rand32(torch.randn, refs.shape, gen=gen)
    ((hs * g_hs).sum() + (refs * g_refs).sum()).backward()
    arrays = dict(tgt=tgt, memory=memory, refpoints_unsigmoid=refu, valid_ratios=valid_ratios, shapes=shapes, lsi=lsi, memory_mask=mmask,
                  attn_mask=amask, hs=hs, refs=refs, grad_hs=g_hs, grad_refs=g_refs, grad_tgt=tgt.grad, grad_memory=memory.grad,
                  grad_refpoints=refu.grad)
    keep = lambda n: n.startswith(("layers.0.", "norm.", "ref_point_head.", "bbox_embed.1."))      # (a 0.9 M-parameter sample of 1.7 M)
    np.savez_compressed(os.path.join(OUT, "decoder256_stack_f64.npz"), **{k: _np(v) for k, v in arrays.items()}, **LP.pack_grads(dec, keep))
    print("decoder256_stack_f64", tuple(hs.shape))


def denoising(dn):
    """prepare_for_cdn (dn_components.py:11-193): the integer / boolean part -- known_bid and map_known_indice are not returned by
    the function, so they are recovered from what it returns: the rows of ``input_query_bbox`` it scatters the boxes into."""
    to_, cuda_ = torch.Tensor.to, torch.Tensor.cuda

    def to_cpu(self, *a, **k):
        a = tuple("cpu" if isinstance(x, str) and x.startswith("cuda") else x for x in a)
        return to_(self, *a, **k)

    torch.Tensor.to, torch.Tensor.cuda = to_cpu, lambda self, *a, **k: self
    try:
        cases = {}
        rng = np.random.default_rng(77)
        for ci, (counts, dn_number, use_cdn, add_gt) in enumerate([
                ((12, 12), 100, True, False), ((3, 7), 100, True, False), ((0, 5), 100, True, False), ((5, 2, 9), 100, False, False),
                ((1, 1), 100, True, True), ((40, 3), 100, True, False), ((2, 6), 10, True, False), ((4, 4), 100, False, True)]):
            targets = []
            for n in counts:
                cxcy = rng.uniform(0.2, 0.8, (n, 2)); wh = rng.uniform(0.05, 0.4, (n, 2))
                targets.append({"labels": torch.as_tensor(rng.integers(1, 1203, n), dtype=torch.long),
                                "boxes": torch.as_tensor(np.concatenate((cxcy, wh), 1), dtype=torch.float32)})
            torch.manual_seed(500 + ci)
            hidden, nq, ncls = 16, 30, 1204
            label_enc = torch.nn.Embedding(ncls + 1, hidden)
            q_label, q_bbox, attn_mask, meta = dn.prepare_for_cdn((targets, dn_number, 0.5, 1.0), True, nq, ncls, hidden, label_enc,
                                                                 use_cdn=use_cdn, add_gt=add_gt)
            pre = f"c{ci}."
            cases[pre + "counts"] = np.asarray(counts, dtype=np.int64)
            cases[pre + "args"] = np.asarray([dn_number, int(use_cdn), int(add_gt), nq], dtype=np.int64)
            cases[pre + "attn_mask"] = attn_mask.numpy()
            cases[pre + "meta"] = np.asarray([meta["pad_size"], meta["num_dn_group"]], dtype=np.int64)
            cases[pre + "query_shape"] = np.asarray(q_bbox.shape, dtype=np.int64)
            # the slots the function filled (every filled box differs from the zero padding: inverse_sigmoid of a box in (0, 1))
            cases[pre + "filled"] = (q_bbox.abs().sum(-1) != 0).numpy()
        np.savez_compressed(os.path.join(OUT, "dn_prepare_for_cdn.npz"), **cases)
        print("dn_prepare_for_cdn:", len(cases) // 6, "cases")
    finally:
        torch.Tensor.to, torch.Tensor.cuda = to_, cuda_


if __name__ == "__main__":
    dt, dn, ut = reference_modules()
    encoder_layer(dt)
    decoder_layer(dt)
    decoder_stack(dt, ut)
    denoising(dn)
    layers_256(dt, ut)
