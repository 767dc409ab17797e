#!/usr/bin/env python3
"""Timing of the small rows around the operator against the reference's op sequences as PyTorch ops:
  * CLIP attention pool on the ROI features of 24 boxes (2048 channels, 7 x 7, 32 heads, 1024 out): the single-query restructuring +
    core kernel against F.multi_head_attention_forward as clip/model.py:73-90 calls it;
  * the matcher's cost blocks for the 7 decoder outputs of a step (bs 2, 900 queries, 1203 classes, 12 boxes per image): one kernel per
    output + one host copy against the reference's op sequence + one .cpu() per output (matcher.py:52-75).

    python tools/time_rows.py
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from richsem_amd.matcher import HungarianMatcher            # noqa: E402
from richsem_amd.modules import AttentionPool2d             # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def main():
    torch.manual_seed(0)
    # ---- attention pool -------------------------------------------------------------------------------------------------------------
    m = AttentionPool2d(7, 2048, 32, 1024).cuda().eval()
    x = torch.randn(int(os.environ.get("ROIS", "24")), 2048, 7, 7, device="cuda")

    def ref_pool(x):
        t = x.flatten(start_dim=2).permute(2, 0, 1)
        t = torch.cat([t.mean(dim=0, keepdim=True), t], dim=0) + m.positional_embedding[:, None, :]
        y, _ = F.multi_head_attention_forward(
            query=t[:1], key=t, value=t, embed_dim_to_check=t.shape[-1], num_heads=32, q_proj_weight=m.q_proj.weight,
            k_proj_weight=m.k_proj.weight, v_proj_weight=m.v_proj.weight, in_proj_weight=None,
            in_proj_bias=torch.cat([m.q_proj.bias, m.k_proj.bias, m.v_proj.bias]), bias_k=None, bias_v=None, add_zero_attn=False,
            dropout_p=0, out_proj_weight=m.c_proj.weight, out_proj_bias=m.c_proj.bias, use_separate_proj_weight=True, training=False,
            need_weights=False)
        return y.squeeze(0)

    with torch.no_grad():
        err = float((m(x) - ref_pool(x)).abs().max() / ref_pool(x).abs().max())
        t1, t2 = timeit(lambda: m(x)), timeit(lambda: ref_pool(x))
    print(f"attention pool, {x.shape[0]} ROIs of 2048 x 7 x 7: restructured + core kernel {t1:.0f} us, F.multi_head_attention_forward {t2:.0f} us "
          f"(max difference {err:.1e} of the output range)")

    # ---- matcher ----------------------------------------------------------------------------------------------------------------------
    bs, nq, C, T = 2, 900, 1203, 12
    outs = [{"pred_logits": torch.randn(bs, nq, C, device="cuda"), "pred_boxes": torch.rand(bs, nq, 4, device="cuda") * 0.5 + 0.25}
            for _ in range(7)]
    targets = [{"labels": torch.randint(0, C, (T,), device="cuda"),
                "boxes": torch.cat([torch.rand(T, 2, device="cuda") * 0.6 + 0.2, torch.rand(T, 2, device="cuda") * 0.3 + 0.05], -1)}
               for _ in range(bs)]
    matcher = HungarianMatcher(2.0, 5.0, 2.0)

    def xyxy(b):
        cx, cy, w, h = b.unbind(-1)
        return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], -1)

    def ref_costs():      # the reference's op sequence up to and including the copy to the host, once per output
        res = []
        tgt_ids = torch.cat([t["labels"] for t in targets])
        tgt_bbox = torch.cat([t["boxes"] for t in targets])
        for o in outs:
            p = o["pred_logits"].flatten(0, 1).sigmoid()
            ob = o["pred_boxes"].flatten(0, 1)
            neg = 0.75 * (p ** 2.0) * (-(1 - p + 1e-8).log())
            pos = 0.25 * ((1 - p) ** 2.0) * (-(p + 1e-8).log())
            cc = pos[:, tgt_ids] - neg[:, tgt_ids]
            cb = torch.cdist(ob, tgt_bbox, p=1)
            b1, b2 = xyxy(ob), xyxy(tgt_bbox)
            a1, a2 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1]), (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
            wh = (torch.min(b1[:, None, 2:], b2[:, 2:]) - torch.max(b1[:, None, :2], b2[:, :2])).clamp(min=0)
            inter = wh[..., 0] * wh[..., 1]
            union = a1[:, None] + a2 - inter
            iou = inter / (union + 1e-6)
            wh2 = (torch.max(b1[:, None, 2:], b2[:, 2:]) - torch.min(b1[:, None, :2], b2[:, :2])).clamp(min=0)
            area = wh2[..., 0] * wh2[..., 1]
            giou = iou - (area - union) / (area + 1e-6)
            res.append((5.0 * cb + 2.0 * cc + 2.0 * (-giou)).view(bs, nq, -1).cpu())
        return res

    from richsem_amd.matcher import CostPlan, cost_blocks

    def ours_costs():     # the library's part of match_many: seven kernels into one buffer, one copy, one wait
        plan = CostPlan(targets, torch.device("cuda"), torch.float32)
        buf = torch.empty(7 * nq * plan.total, device="cuda")
        for i, o in enumerate(outs):
            cost_blocks(o["pred_logits"], o["pred_boxes"], plan, 2.0, 5.0, 2.0, 0.25, out=buf[i * nq * plan.total:(i + 1) * nq * plan.total])
        host = matcher._host_buffer(buf.numel(), torch.float32)
        host.copy_(buf, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return host

    t3, t4 = timeit(ours_costs, 10), timeit(ref_costs, 10)
    t5 = timeit(lambda: matcher.match_many(outs, targets), 5)
    print(f"matching costs of 7 decoder outputs on the host: library {t3:.0f} us, reference op sequence with a .cpu() per output {t4:.0f} us; "
          f"match_many incl. scipy's assignments {t5:.0f} us")


if __name__ == "__main__":
    main()
