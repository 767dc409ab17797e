#!/usr/bin/env python3
"""Golden vectors for SURVEY.md section 8f rank 2 (the two-stage class score), generated from the REFERENCE's own ``CLIPAlign.forward``.

Run in the build container only (it reads /root/reference; the fixture it writes is committed, the GPU box never sees the reference):

    python tests/golden/make_golden_cls.py

``models/richsem/richsem.py`` cannot be imported here (clip / torchvision / detectron2 at module level), so the class ``CLIPAlign``
(richsem.py:38-205) is cut out of the source with ``ast`` and executed on its own.  Its ``__init__`` builds a CLIP model (weights that are
not in the image), so the instance is made without it and given exactly the attributes ``forward`` reads in the shipped configuration
(richsem.py:75-83, 176-184): ``dino_visual_proj`` = a bias-free ``nn.Linear``, ``text_proj = None``, ``text_embed``, ``logit_scale`` --
seeded values.  What runs is the reference's ``forward`` / ``_get_text_features``; the two torch calls that follow it in
``deformable_transformer.py:368-372`` (``.max(-1)[0]`` and ``torch.topk(..., dim=1)[1]``) are applied to its result here.
"""
import ast
import copy
import os

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def reference_class():
    path = f"{REF}/models/richsem/richsem.py"
    tree = ast.parse(open(path).read())
    body = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "CLIPAlign"]
    assert len(body) == 1
    ns = {"torch": torch, "nn": nn, "F": F, "copy": copy}
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return ns["CLIPAlign"]


def main():
    CLIPAlign = reference_class()
    out = {}
    for tag, dtype, seed, bs, S, classes, proj, k in (("f64", torch.float64, 5, 2, 120, 97, 48, 20), ("f32", torch.float32, 6, 2, 300, 1204, 64, 60)):
        g = torch.Generator().manual_seed(seed)
        m = CLIPAlign.__new__(CLIPAlign)
        nn.Module.__init__(m)
        m.dino_visual_proj = nn.Linear(256, proj, bias=False).to(dtype)
        with torch.no_grad():
            m.dino_visual_proj.weight.copy_(torch.randn(proj, 256, generator=g, dtype=dtype) * proj ** -0.5)
        m.text_proj = None
        m.text_embed = torch.randn(classes, proj, generator=g, dtype=dtype)
        m.logit_scale = nn.Parameter(torch.tensor(np.log(1 / 0.07), dtype=dtype), requires_grad=False)
        memory = torch.randn(bs, S, 256, generator=g, dtype=dtype)
        with torch.no_grad():
            logits = m(memory)                                        # the reference's CLIPAlign.forward
            scores = logits.max(-1)[0]                                # deformable_transformer.py:371
            topk = torch.topk(scores, k, dim=1)[1]                    # :371
        out.update({f"{tag}.memory": memory.numpy(), f"{tag}.proj_weight": m.dino_visual_proj.weight.detach().numpy(),
                    f"{tag}.text_embed": m.text_embed.numpy(), f"{tag}.logit_scale": m.logit_scale.detach().numpy(),
                    f"{tag}.scores": scores.numpy(), f"{tag}.topk": topk.numpy(),
                    f"{tag}.logits_head": logits[:, :8].numpy()})    # (a slice of the full logits: the whole tensor is not needed)
        print(tag, tuple(logits.shape), tuple(topk.shape))
    np.savez_compressed(os.path.join(OUT, "cls_clipalign.npz"), **out)


if __name__ == "__main__":
    main()
