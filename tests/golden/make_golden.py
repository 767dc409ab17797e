#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE's own implementation.

Run in the build container only (needs /root/reference; it does not exist on the GPU box):

    python tests/golden/make_golden.py

What is executed from the reference: the pure-PyTorch function ``ms_deform_attn_core_pytorch``
(reference models/richsem/ops/functions/ms_deform_attn_func.py:41-61), loaded by file path, and
torch autograd through it in fp64 (the backward oracle).  That file does
``import MultiScaleDeformableAttention as MSDA`` on line 18 (the compiled CUDA extension, which
cannot be built here); the CPU function never touches it, so an empty module object is
registered under that name for the duration of the import, as SURVEY.md section 8(c) records.

Input recipes follow the reference's only test, models/richsem/ops/test.py:
  * shapes / sizes            test.py:21-25   (N,M,D = 1,2,2; Lq,L,P = 2,2,2; shapes (6,4),(3,2))
  * seed                      test.py:28      torch.manual_seed(3); the draws are made on the CPU
                                              generator (``torch.rand(..).cuda()``), so they
                                              reproduce here bit for bit
  * value/loc/attn recipe     test.py:33-36
  * order of draws            test.py:81-86   double fwd, float fwd, then one draw per channel
                                              count of the gradient check
Only data is written (inputs and expected outputs, .npz); no reference source is copied.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/models/richsem/ops/functions/ms_deform_attn_func.py"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference_core():
    name = "MultiScaleDeformableAttention"
    had = sys.modules.get(name)
    sys.modules[name] = types.ModuleType(name)  # never called by the CPU function
    try:
        spec = importlib.util.spec_from_file_location("_ref_ms_deform_attn_func", REF)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        if had is None:
            del sys.modules[name]
        else:
            sys.modules[name] = had
    return mod.ms_deform_attn_core_pytorch


def lsi_of(shapes):
    return torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))


def run_case(core, value, shapes, loc, aw, grad_out, dtype):
    """Forward + autograd grads through the reference function, in `dtype`."""
    v = value.to(dtype).clone().requires_grad_(True)
    l = loc.to(dtype).clone().requires_grad_(True)
    a = aw.to(dtype).clone().requires_grad_(True)
    out = core(v, shapes, l, a)
    out.backward(grad_out.to(dtype))
    return out.detach(), v.grad, l.grad, a.grad


def save(name, dtype, value, shapes, loc, aw, grad_out, res):
    npdt = {torch.float64: np.float64, torch.float32: np.float32}[dtype]
    out, gv, gl, ga = res
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        value=value.numpy().astype(npdt), shapes=shapes.numpy(), lsi=lsi_of(shapes).numpy(),
        loc=loc.numpy().astype(npdt), aw=aw.numpy().astype(npdt),
        grad_out=grad_out.numpy().astype(npdt),
        out=out.numpy(), grad_value=gv.numpy(), grad_loc=gl.numpy(), grad_aw=ga.numpy())
    print(f"{name}: value{tuple(value.shape)} loc{tuple(loc.shape)} -> out{tuple(out.shape)}")


def recipe(N, S, M, D, Lq, L, P):
    """test.py:33-36"""
    value = torch.rand(N, S, M, D) * 0.01
    loc = torch.rand(N, Lq, M, L, P, 2)
    aw = torch.rand(N, Lq, M, L, P) + 1e-5
    aw /= aw.sum(-1, keepdim=True).sum(-2, keepdim=True)
    return value, loc, aw


def main():
    core = load_reference_core()

    # ---- 1. the reference test's own sequence (test.py:81-86) ---------------------------------
    N, M, D = 1, 2, 2
    Lq, L, P = 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    torch.manual_seed(3)
    g = torch.Generator().manual_seed(1234)  # grad_output: not part of the reference recipe

    value, loc, aw = recipe(N, S, M, D, Lq, L, P)              # check_forward_..._double
    go = torch.randn(N, Lq, M * D, generator=g, dtype=torch.float64)
    save("ref_test_fwd_double", torch.float64, value, shapes, loc, aw, go,
         run_case(core, value, shapes, loc, aw, go, torch.float64))

    value, loc, aw = recipe(N, S, M, D, Lq, L, P)              # check_forward_..._float
    go = torch.randn(N, Lq, M * D, generator=g, dtype=torch.float64)
    save("ref_test_fwd_float", torch.float32, value, shapes, loc, aw, go,
         run_case(core, value, shapes, loc, aw, go, torch.float32))

    for ch in [30, 32, 64, 71, 1025, 2048, 3096]:             # check_gradient_numerical(channels)
        value, loc, aw = recipe(N, S, M, ch, Lq, L, P)
        go = torch.randn(N, Lq, M * ch, generator=g, dtype=torch.float64)
        if ch <= 71:                                           # larger ones: draws only, to keep
            save(f"ref_test_grad_D{ch}", torch.float64, value, shapes, loc, aw, go,   # the stream
                 run_case(core, value, shapes, loc, aw, go, torch.float64))

    # ---- 2. border / out-of-range sampling locations -----------------------------------------
    # power-of-two maps so that h_im = -1 and h_im = H are hit exactly in binary floating point
    shapes = torch.as_tensor([(8, 4), (4, 2)], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    N, M, D, L, P = 1, 2, 4, 2, 8
    gen = torch.Generator().manual_seed(77)
    value = torch.randn(N, S, M, D, generator=gen, dtype=torch.float64)
    # forward-only set: includes the exact acceptance-window borders (the CUDA kernel skips the
    # sample there, grid_sample gives it zero weight: same output, different one-sided gradient)
    ys = torch.tensor([-0.5 / 8, (8 + 0.5) / 8, -0.3, 1.4, 0.0, 1.0, 0.5 / 8, (8 - 0.5) / 8],
                      dtype=torch.float64)
    xs = torch.tensor([-0.5 / 4, (4 + 0.5) / 4, 0.0, 1.0, -0.2, 1.3, 0.5 / 4, (4 - 0.5) / 4],
                      dtype=torch.float64)
    Lq = 8
    loc = torch.empty(N, Lq, M, L, P, 2, dtype=torch.float64)
    for q in range(Lq):
        for p in range(P):
            loc[:, q, :, :, p, 0] = xs[(p + q) % 8]
            loc[:, q, :, :, p, 1] = ys[p]
    loc[:, :, 1] = loc[:, :, 1].flip(-1)          # head 1: swap the roles of x and y
    aw = torch.rand(N, Lq, M, L, P, generator=gen, dtype=torch.float64) + 1e-5
    aw /= aw.sum((-1, -2), keepdim=True)
    go = torch.randn(N, Lq, M * D, generator=gen, dtype=torch.float64)
    res = run_case(core, value, shapes, loc, aw, go, torch.float64)
    save("border_exact_fwd", torch.float64, value, shapes, loc, aw, go, res)

    # gradient set: locations in the partially-outside bands (-1,0) and (H-1,H), far outside,
    # and inside -- but never exactly on a border or on an integer pixel coordinate
    Lq = 16
    loc = torch.rand(N, Lq, M, L, P, 2, generator=gen, dtype=torch.float64) * 1.6 - 0.3
    aw = torch.rand(N, Lq, M, L, P, generator=gen, dtype=torch.float64) + 1e-5
    aw /= aw.sum((-1, -2), keepdim=True)
    go = torch.randn(N, Lq, M * D, generator=gen, dtype=torch.float64)
    save("border_bands_grad", torch.float64, value, shapes, loc, aw, go,
         run_case(core, value, shapes, loc, aw, go, torch.float64))

    # ---- 3. shrunken 4-level pyramid with the real aspect ratio, encoder-like (Lq == S) -------
    shapes = torch.as_tensor([(13, 21), (7, 11), (4, 6), (2, 3)], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    N, M, D, L, P = 1, 4, 32, 4, 4
    gen = torch.Generator().manual_seed(5)
    value = torch.randn(N, S, M, D, generator=gen, dtype=torch.float64)
    # reference points = pixel centres of every level (deformable_transformer.py:512-525 with
    # valid_ratio 1), offsets = the module's initial bias pattern (ms_deform_attn.py:62-76) + jitter
    refs = []
    for H, W in shapes.tolist():
        ry, rx = torch.meshgrid(torch.linspace(0.5, H - 0.5, H, dtype=torch.float64),
                                torch.linspace(0.5, W - 0.5, W, dtype=torch.float64), indexing="ij")
        refs.append(torch.stack((rx.reshape(-1) / W, ry.reshape(-1) / H), -1))
    ref = torch.cat(refs, 0)                                            # (S, 2) in (x, y)
    th = torch.arange(M, dtype=torch.float64) * (2.0 * torch.pi / M)
    d = torch.stack([th.cos(), th.sin()], -1)
    d = d / d.abs().max(-1, keepdim=True)[0]                            # (M, 2)
    k = torch.arange(1, P + 1, dtype=torch.float64)
    off = d[:, None, None, :] * k[None, None, :, None]                  # (M, 1, P, 2) pixels
    off = off.expand(M, L, P, 2) + torch.randn(N, S, M, L, P, 2, generator=gen, dtype=torch.float64)
    norm = torch.stack([shapes[:, 1], shapes[:, 0]], -1).to(torch.float64)   # (L, 2) = (W, H)
    loc = ref[None, :, None, None, None, :] + off / norm[None, None, None, :, None, :]
    aw = torch.softmax(torch.randn(N, S, M, L * P, generator=gen, dtype=torch.float64), -1)
    aw = aw.view(N, S, M, L, P)
    go = torch.randn(N, S, M * D, generator=gen, dtype=torch.float64)
    save("pyramid_encoder_f64", torch.float64, value, shapes, loc, aw, go,
         run_case(core, value, shapes, loc, aw, go, torch.float64))
    save("pyramid_encoder_f32", torch.float32, value, shapes, loc, aw, go,
         run_case(core, value, shapes, loc, aw, go, torch.float32))

    # ---- 4. two images x eight heads (the (batch,head) -> XCD mapping), decoder-like Lq != S ---
    shapes = torch.as_tensor([(6, 10), (3, 5)], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    N, M, D, L, P, Lq = 2, 8, 32, 2, 4, 19
    gen = torch.Generator().manual_seed(11)
    value = torch.randn(N, S, M, D, generator=gen, dtype=torch.float64)
    loc = torch.rand(N, Lq, M, L, P, 2, generator=gen, dtype=torch.float64) * 1.2 - 0.1
    aw = torch.softmax(torch.randn(N, Lq, M, L * P, generator=gen, dtype=torch.float64), -1)
    aw = aw.view(N, Lq, M, L, P)
    go = torch.randn(N, Lq, M * D, generator=gen, dtype=torch.float64)
    save("decoder_n2m8_f64", torch.float64, value, shapes, loc, aw, go,
         run_case(core, value, shapes, loc, aw, go, torch.float64))


if __name__ == "__main__":
    main()


# ---------------------------------------------------------------------------------------------------------------
# Module-level vectors: the reference's own nn.Module
# (models/richsem/ops/modules/ms_deform_attn.py:30-115) run on the CPU.  The module file does
# `from ..functions import MSDeformAttnFunction`, i.e. it needs its package and the compiled CUDA function; here the
# module source file is loaded by path into a synthetic package whose `functions.MSDeformAttnFunction.apply`
# forwards to the reference's own pure-PyTorch core (the function its authors ship "for debug and test"), so every
# arithmetic step that produces the vectors is the reference's.
def make_module_vectors():
    core = load_reference_core()
    pkg = types.ModuleType("_refops")
    pkg.__path__ = []
    fn_mod = types.ModuleType("_refops.functions")

    class MSDeformAttnFunction:   # same call signature as the reference's autograd Function (ms_deform_attn_func.py:23)
        @staticmethod
        def apply(value, shapes, lsi, loc, aw, im2col_step):
            return core(value, shapes, loc, aw)

    fn_mod.MSDeformAttnFunction = MSDeformAttnFunction
    mods = types.ModuleType("_refops.modules")
    mods.__path__ = []
    sys.modules.update({"_refops": pkg, "_refops.functions": fn_mod, "_refops.modules": mods})
    path = "/root/reference/models/richsem/ops/modules/ms_deform_attn.py"
    spec = importlib.util.spec_from_file_location("_refops.modules.ms_deform_attn", path)
    m = importlib.util.module_from_spec(spec)
    m.__package__ = "_refops.modules"
    spec.loader.exec_module(m)
    RefMSDeformAttn = m.MSDeformAttn

    shapes = torch.as_tensor([(9, 14), (5, 7), (3, 4), (2, 2)], dtype=torch.long)
    lsi = lsi_of(shapes)
    S = int(shapes.prod(1).sum())
    N, C, L, H, P = 2, 64, 4, 2, 4     # two heads of 32 channels: small fixture, same per-head width as RichSem
    for name, ref_dim, Lq in (("module_encoder_ref2d", 2, S), ("module_decoder_ref4d", 4, 37)):
        torch.manual_seed(20 if ref_dim == 2 else 21)
        mod = RefMSDeformAttn(C, L, H, P).double()
        with torch.no_grad():   # move away from the all-zero init so every parameter matters
            mod.sampling_offsets.weight.normal_(0, 0.02)
            mod.attention_weights.weight.normal_(0, 0.05)
            mod.attention_weights.bias.normal_(0, 0.1)
        gen = torch.Generator().manual_seed(99 + ref_dim)
        query = torch.randn(N, Lq, C, generator=gen, dtype=torch.float64, requires_grad=True)
        src = torch.randn(N, S, C, generator=gen, dtype=torch.float64, requires_grad=True)
        if ref_dim == 2:   # pixel centres, as TransformerEncoder.get_reference_points with valid_ratio 1
            refs = []
            for Hh, Ww in shapes.tolist():
                ry, rx = torch.meshgrid(torch.linspace(0.5, Hh - 0.5, Hh, dtype=torch.float64),
                                        torch.linspace(0.5, Ww - 0.5, Ww, dtype=torch.float64), indexing="ij")
                refs.append(torch.stack((rx.reshape(-1) / Ww, ry.reshape(-1) / Hh), -1))
            rp = torch.cat(refs, 0)[None, :, None, :].expand(N, S, L, 2).contiguous()
        else:              # boxes (cx, cy, w, h)
            cxcy = torch.rand(N, Lq, 1, 2, generator=gen, dtype=torch.float64) * 0.8 + 0.1
            wh = torch.rand(N, Lq, 1, 2, generator=gen, dtype=torch.float64) * 0.4 + 0.05
            rp = torch.cat((cxcy, wh), -1).expand(N, Lq, L, 4).contiguous()
        mask = torch.zeros(N, S, dtype=torch.bool)
        mask[1, lsi[0] + 9:lsi[0] + 14] = True          # padded pixels of image 1
        mask[1, -1] = True
        out = mod(query, rp, src, shapes, lsi, mask)
        go = torch.randn(out.shape, generator=gen, dtype=torch.float64)
        out.backward(go)
        sd = {k: v.detach().numpy() for k, v in mod.state_dict().items()}
        grads = {k + ".grad": p.grad.numpy() for k, p in mod.named_parameters()}
        np.savez_compressed(os.path.join(OUT, name + ".npz"), query=query.detach().numpy(), src=src.detach().numpy(),
                            reference_points=rp.numpy(), shapes=shapes.numpy(), lsi=lsi.numpy(), mask=mask.numpy(),
                            out=out.detach().numpy(), grad_out=go.numpy(), grad_query=query.grad.numpy(),
                            grad_src=src.grad.numpy(), **{"param." + k: v for k, v in sd.items()},
                            **{"param." + k: v for k, v in grads.items()})
        print(f"{name}: query{tuple(query.shape)} src{tuple(src.shape)} ref{tuple(rp.shape)} -> out{tuple(out.shape)}")


if __name__ == "__main__":
    make_module_vectors()
