"""GPU (-m gpu): the rows chained the way RichSem chains them (models/richsem/richsem.py:593-640, deformable_transformer.py:720-823,
:368-380): backbone -> input projections -> encoder layer -> two-stage query selection -> decoder layer, plus the frozen CLIP teacher ->
ROIAlign -> attention pool -> text logits and the matcher on the decoder's outputs -- every stage on the library's kernels, checked for
shapes, finiteness and (where an oracle exists for the composition) values.  A smoke test of the interfaces between the rows, at a small
image size; each row has its own parity tests."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu


def test_rows_compose():
    from richsem_amd.backbone import InputProj, ResNet50Frozen
    from richsem_amd.clip_resnet import ModifiedResNetTeacher
    from richsem_amd.matcher import HungarianMatcher
    from richsem_amd.modules import (DeformableTransformerDecoderLayer, DeformableTransformerEncoderLayer, clip_box_targets,
                                     get_reference_points)
    from richsem_amd.two_stage import ClassScorer
    from richsem_amd.workload import resnet50_state_dict
    from clip_resnet_params import fill_state_dict
    from test_gpu_backbone import input_proj_state_dict
    from test_oracle_clip_resnet import template_state_dict

    torch.manual_seed(0)
    dev = "cuda"
    N, H, W = 2, 128, 192
    images = torch.randn(N, 3, H, W, device=dev)

    # backbone (layers reduced to one block each) + input projections: four levels of 256-wide tokens
    stages = ResNet50Frozen(resnet50_state_dict(seed=1, layers=(1, 1, 1, 1)))(images)
    srcs, shapes = InputProj(input_proj_state_dict())(stages)
    assert shapes == [(16, 24), (8, 12), (4, 6), (2, 3)] and all(s.shape == (N, h * w, 256) for s, (h, w) in zip(srcs, shapes))
    memory = torch.cat(srcs, 1)                                                     # (N, S, 256)
    S = memory.shape[1]
    spatial = torch.tensor(shapes, dtype=torch.int64, device=dev)
    lsi = torch.cat((spatial.new_zeros(1), spatial.prod(1).cumsum(0)[:-1]))
    assert torch.isfinite(memory).all()

    # one encoder layer in bf16 (fused attention module + one-kernel feed-forward block)
    enc = DeformableTransformerEncoderLayer(256, 2048, dropout=0.0, n_levels=4, n_heads=8, n_points=4).to(dev)
    enc.fused_min_tokens = 0
    ref = get_reference_points(shapes, torch.ones(N, 4, 2, device=dev), dev)
    mem16 = enc(memory.to(torch.bfloat16), None, ref, spatial, lsi, None)
    assert mem16.shape == (N, S, 256) and mem16.dtype == torch.bfloat16 and torch.isfinite(mem16.float()).all()

    # two-stage selection: the 20 tokens with the largest class score (no logit tensor)
    classes, proj = 50, 64
    wp, text = torch.randn(proj, 256, device=dev) * proj ** -0.5, torch.randn(classes, proj, device=dev)
    scorer = ClassScorer(2).prepare(wp, text, torch.tensor(2.0))
    topk = scorer.topk_proposals(mem16.float(), 20)
    f = mem16.float() @ wp.t()
    logits = np.exp(2.0) * (f / f.norm(dim=-1, keepdim=True)) @ (text / text.norm(dim=-1, keepdim=True)).t()
    want = torch.topk(logits.max(-1)[0], 20, dim=1)[1]
    assert topk.shape == (N, 20) and sum(len(set(a.tolist()) ^ set(b.tolist())) for a, b in zip(topk, want)) <= 2

    # one decoder layer on the selected queries (4-d reference boxes), fp32
    dec = DeformableTransformerDecoderLayer(256, 2048, dropout=0.0, n_levels=4, n_heads=8, n_points=4).to(dev)
    tgt = torch.gather(mem16.float(), 1, topk[..., None].expand(-1, -1, 256)).transpose(0, 1).contiguous()          # (nq, N, 256)
    boxes = torch.rand(20, N, 4, device=dev) * 0.4 + 0.3
    refp = boxes[:, :, None, :].expand(-1, -1, 4, -1).contiguous()
    hs = dec(tgt, None, None, None, refp, mem16.float().transpose(0, 1).contiguous(), None, lsi, spatial)
    assert hs.shape == (20, N, 256) and torch.isfinite(hs).all()

    # matcher on the decoder's outputs (class logits through the same classifier, boxes as they are)
    out_logits = (hs.transpose(0, 1) @ wp.t()) @ text.t()
    targets = [{"labels": torch.randint(0, classes, (3,), device=dev), "boxes": torch.rand(3, 4, device=dev) * 0.3 + 0.3,
                "size": torch.tensor([float(H), float(W)], device=dev)} for _ in range(N)]
    idx = HungarianMatcher(2.0, 5.0, 2.0)({"pred_logits": out_logits, "pred_boxes": boxes.transpose(0, 1).contiguous()}, targets)
    assert len(idx) == N and all(len(i) == 3 and len(set(i.tolist())) == 3 for i, _ in idx)

    # the frozen teacher: feature map -> ROIAlign of the ground-truth boxes -> attention pool -> text logits
    sd = fill_state_dict(template_state_dict((1, 1, 1, 1), 64, 32, 64, 224), 5)
    teacher = ModifiedResNetTeacher(sd, heads=32)
    _, fmap = teacher(images, ret_sp=True)
    assert fmap.shape == (N, 2048, H // 32, W // 32)
    prompts, tlogits = clip_box_targets(fmap, targets, teacher.attnpool, torch.randn(classes, 64, device=dev), 2.0)
    assert [p.shape for p in prompts] == [(3, 64)] * N and [t.shape for t in tlogits] == [(3, classes)] * N
    assert all(torch.isfinite(t).all() for t in tlogits)
    assert "librichsem_msda.so" in open("/proc/self/maps").read()
