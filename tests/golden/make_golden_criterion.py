#!/usr/bin/env python3
"""Golden vectors for the criterion kernels of the composed step (``msda_focal_neg_*``, ``msda_focal_pos_sum_f32``,
``msda_box_pair_loss_f32``; richsem_amd/matcher.py: FocalNegativeSum, FocalPositiveSum, BoxPairLoss), generated from the REFERENCE's own
loss functions.  Run in the build container only (it reads /root/reference; the fixture is committed, the GPU box never sees the reference):

    python tests/golden/make_golden_criterion.py

What is executed is the reference's code:
  * ``sigmoid_focal_loss`` of ``models/richsem/utils.py:82-108`` (the file is loaded by path: it needs torch only), called as
    ``SetCriterion.loss_labels`` calls it (richsem.py:938-964: class targets scattered into a one-hot tensor with a no-object column that is
    cut off again, ``* src_logits.shape[1]``), for a matching-part shaped and a denoising-part shaped output;
  * ``box_cxcywh_to_xyxy`` / ``box_iou`` / ``generalized_box_iou`` of ``util/box_ops.py:9-64``, cut out of the source with ``ast`` (the file
    imports ``torchvision.ops.boxes.box_area``; torchvision is absent, that ONE function is restated from its published definition, as in
    make_golden_matcher.py), called as ``SetCriterion.loss_boxes`` calls them (richsem.py: ``F.l1_loss(..., reduction='none')``,
    ``1 - torch.diag(generalized_box_iou(...))``).
Everything runs in float64 on inputs that are exactly representable in float32 (the kernels compute in fp32); gradients by autograd.
"""
import ast
import importlib.util
import os

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def box_area(boxes):      # torchvision.ops.boxes.box_area (published definition; torchvision is absent from the image)
    return (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])


def reference_box_ops():
    names = ["box_cxcywh_to_xyxy", "box_iou", "generalized_box_iou"]
    path = f"{REF}/util/box_ops.py"
    tree = ast.parse(open(path).read())
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(body) == len(names)
    ns = {"torch": torch, "box_area": box_area}
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return ns


def reference_utils():
    spec = importlib.util.spec_from_file_location("_ref_utils", f"{REF}/models/richsem/utils.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def f32exact(t):
    return t.float().double()


def main():
    utils, ops = reference_utils(), reference_box_ops()
    out = {}
    # ---- classification: loss_labels' call of sigmoid_focal_loss (richsem.py:938-964) --------------------------------------------------
    for tag, seed, N, Q, C, n_pos, num_boxes in (("match", 21, 2, 90, 57, 11, 22.0), ("dn", 22, 2, 48, 401, 24, 240.0)):
        g = torch.Generator().manual_seed(seed)
        logits = f32exact(torch.randn(N, Q, C, generator=g) * 4)
        logits[0, 0, :6] = torch.tensor([25.0, -25.0, 0.0, 19.999, 20.001, -60.0], dtype=torch.float64)   # softplus's branches, saturation
        target_classes = torch.full((N, Q), C, dtype=torch.int64)
        for b in range(N):
            q = torch.randperm(Q, generator=g)[:n_pos]
            target_classes[b, q] = torch.randint(0, C, (n_pos,), generator=g)
        target_classes[0, 0] = 3                                         # a positive at a saturated logit row
        x = logits.clone().requires_grad_(True)
        onehot = torch.zeros((N, Q, C + 1), dtype=x.dtype)
        onehot.scatter_(2, target_classes.unsqueeze(-1), 1)
        onehot = onehot[:, :, :-1]
        loss = utils.sigmoid_focal_loss(x, onehot, num_boxes, alpha=0.25, gamma=2) * Q
        loss.backward()
        out[f"focal_{tag}.logits"] = logits.float().numpy()
        out[f"focal_{tag}.target_classes"] = target_classes.numpy()
        out[f"focal_{tag}.num_boxes"] = np.float64(num_boxes)
        out[f"focal_{tag}.loss"] = loss.detach().numpy()
        out[f"focal_{tag}.grad"] = x.grad.numpy()
        print(tag, float(loss))
    # ---- boxes: loss_boxes' L1 and GIoU of matched pairs --------------------------------------------------------------------------------
    g = torch.Generator().manual_seed(23)
    K = 257
    tb = f32exact(torch.cat((torch.rand(K, 2, generator=g) * 0.6 + 0.2, torch.rand(K, 2, generator=g) * 0.3 + 0.02), -1))
    pb = f32exact((tb + 0.1 * torch.randn(K, 4, generator=g).double()).clamp(0.01, 0.99))
    pb[:5] = tb[:5]                                                       # identical boxes (ties in every max / min)
    pb[5:10, :2] = f32exact(tb[5:10, :2] + 0.45)                          # mostly disjoint
    pb[10:15, :2] = tb[10:15, :2]
    pb[10:15, 2:] = f32exact(tb[10:15, 2:] * 0.25)                        # nested
    w = f32exact(torch.rand(K, generator=g))
    w[::17] = 0.0
    x = pb.clone().requires_grad_(True)
    l1 = F.l1_loss(x, tb, reduction="none").sum(-1)
    giou = torch.diag(ops["generalized_box_iou"](ops["box_cxcywh_to_xyxy"](x), ops["box_cxcywh_to_xyxy"](tb)))
    total = ((5.0 * l1 + 2.0 * (1 - giou)) * w).sum()
    total.backward()
    out["box.pred"], out["box.tgt"], out["box.w"] = pb.float().numpy(), tb.float().numpy(), w.float().numpy()
    out["box.l1"], out["box.giou"] = l1.detach().numpy(), giou.detach().numpy()
    out["box.loss"], out["box.grad"] = total.detach().numpy(), x.grad.numpy()
    print("box", float(total))
    np.savez_compressed(os.path.join(OUT, "criterion_reference.npz"), **out)


if __name__ == "__main__":
    main()
