"""TEST INFRASTRUCTURE (oracle) -- numpy restatement of the reference's two-stage class score, never imported by the product.

Follows models/richsem/richsem.py:176-184 (the classifier's forward with a bias-free linear ``dino_visual_proj``, :75-83) and
models/richsem/deformable_transformer.py:368-372 (row maximum, top-k), written the plain way: project, normalise, multiply with the
normalised text embeddings, scale, take the maximum.  **PINNED** (round 3): tests/golden/cls_clipalign.npz was produced by the reference's
own ``CLIPAlign.forward`` (the class cut out of richsem.py with ``ast`` -- the module imports clip / torchvision / detectron2, absent from
the image -- and given seeded ``dino_visual_proj`` / ``text_embed`` / ``logit_scale``; tests/golden/make_golden_cls.py) followed by the
reference's ``.max(-1)[0]`` and ``torch.topk``; tests/test_oracle_cls.py holds this file to it (1e-12 in fp64, 2e-5 in fp32, identical
selection)."""
import numpy as np


def class_logits(memory, proj_weight, text_embed, logit_scale):
    """memory (..., 256) -> (..., classes), in memory's dtype"""
    f = memory @ proj_weight.T                                             # richsem.py:178
    f = f / np.linalg.norm(f, axis=-1, keepdims=True)                      # :179
    t = text_embed / np.linalg.norm(text_embed, axis=-1, keepdims=True)    # :180
    return np.exp(logit_scale).astype(memory.dtype) * (f @ t.T)            # :181-182


def max_logits(memory, proj_weight, text_embed, logit_scale):
    return class_logits(memory, proj_weight, text_embed, logit_scale).max(axis=-1)      # deformable_transformer.py:371


def topk(scores, k):
    """torch.topk(scores, k, dim=1)[1] with ties lowest index first"""
    return np.argsort(-scores, axis=1, kind="stable")[:, :k]
