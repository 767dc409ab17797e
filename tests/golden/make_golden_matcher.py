#!/usr/bin/env python3
"""Golden vectors for SURVEY.md section 8f rank 4 (the matcher's cost blocks), generated from the REFERENCE's own ``HungarianMatcher``.

Run in the build container only (it reads /root/reference; the fixture it writes is committed, the GPU box never sees the reference):

    python tests/golden/make_golden_matcher.py

What is executed is the reference's code: ``models/richsem/matcher.py`` is loaded by file path; its one import of the reference tree,
``util.box_ops`` (util/box_ops.py:9-59), is served by that file's own ``box_cxcywh_to_xyxy``, ``box_iou`` and ``generalized_box_iou``, cut
out of the source with ``ast`` and executed -- the file as a whole imports ``torchvision.ops.boxes.box_area``, and torchvision
(``torchvision>=0.6.0``, requirements.txt:5) is not in the image.  That ONE third-party function is restated here from its published
definition -- ``(boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])`` -- and handed to the executed reference functions; nothing else
is ours.  The cost blocks are what the reference passes to ``scipy.optimize.linear_sum_assignment`` (matcher.py:76-77), recorded on the
way; the indices are what its ``forward`` returns.
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


def box_area(boxes):      # torchvision.ops.boxes.box_area (published definition; torchvision is absent from the image)
    return (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])


def reference_matcher_module(recorded):
    names = ["box_cxcywh_to_xyxy", "box_iou", "generalized_box_iou"]
    path = f"{REF}/util/box_ops.py"
    tree = ast.parse(open(path).read())
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(body) == len(names)
    ns = {"torch": torch, "box_area": box_area}
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    util = types.ModuleType("util"); util.__path__ = []
    box_ops = types.ModuleType("util.box_ops")
    for n in names:
        setattr(box_ops, n, ns[n])
    util.box_ops = box_ops
    sys.modules.update({"util": util, "util.box_ops": box_ops})
    spec = importlib.util.spec_from_file_location("_ref_matcher", f"{REF}/models/richsem/matcher.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    real = m.linear_sum_assignment

    def recording(c):      # the block the reference hands to scipy, and scipy's answer
        recorded.append(c.detach().clone().numpy())
        return real(c)

    m.linear_sum_assignment = recording
    return m


def main():
    recorded = []
    m = reference_matcher_module(recorded)
    out = {}
    for tag, dtype, seed, bs, nq, ncls, sizes in (("f64", torch.float64, 11, 3, 40, 57, (5, 0, 9)), ("f32", torch.float32, 12, 2, 120, 1203, (12, 7))):
        g = torch.Generator().manual_seed(seed)
        logits = torch.randn(bs, nq, ncls, generator=g, dtype=dtype) * 2
        cxcy = torch.rand(bs, nq, 2, generator=g, dtype=dtype) * 0.8 + 0.1
        wh = torch.rand(bs, nq, 2, generator=g, dtype=dtype) * 0.3 + 0.01
        boxes = torch.cat((cxcy, wh), -1)
        targets = []
        for k in sizes:
            tc = torch.rand(k, 2, generator=g, dtype=dtype) * 0.8 + 0.1
            tw = torch.rand(k, 2, generator=g, dtype=dtype) * 0.3 + 0.01
            targets.append({"labels": torch.randint(0, ncls, (k,), generator=g), "boxes": torch.cat((tc, tw), -1)})
        del recorded[:]
        matcher = m.HungarianMatcher(cost_class=2.0, cost_bbox=5.0, cost_giou=2.0)      # the shipped weights (config: set_cost_*)
        idx = matcher(dict(pred_logits=logits, pred_boxes=boxes), targets)
        assert len(recorded) == bs
        out[f"{tag}.logits"], out[f"{tag}.boxes"] = logits.numpy(), boxes.numpy()
        out[f"{tag}.sizes"] = np.asarray(sizes, dtype=np.int64)
        out[f"{tag}.labels"] = torch.cat([t["labels"] for t in targets]).numpy()
        out[f"{tag}.tgt_boxes"] = torch.cat([t["boxes"] for t in targets]).numpy()
        for b in range(bs):
            out[f"{tag}.block{b}"] = recorded[b]
            out[f"{tag}.idx_i{b}"], out[f"{tag}.idx_j{b}"] = idx[b][0].numpy(), idx[b][1].numpy()
        print(tag, [r.shape for r in recorded])
    np.savez_compressed(os.path.join(OUT, "matcher_hungarian.npz"), **out)


if __name__ == "__main__":
    main()
