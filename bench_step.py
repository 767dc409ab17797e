"""A composed RichSem training step on richsem_amd's rows, for bench.py's ``full_step`` line.

NOT a model framework: the tier this repository is built to covers the operator and the rows around it, not RichSem's model assembly
(out of scope).  This file chains those rows the way the reference chains them --
``engine.py:44-114`` -> ``models/richsem/richsem.py:581-774`` -> ``deformable_transformer.py:273-463`` -- at the tensor sizes of
BASELINE.json configs[1] (2 x 800 x 1333 images padded to 1344, R50 4-scale, 900 queries + denoising groups of 12 boxes, 1204 classes,
frozen CLIP-RN50 teacher), with synthetic seeded weights and a compact criterion (the reference's SetCriterion stays host Python by
``north_star``; what is here reproduces its tensor work: Hungarian matching of 6 decoder outputs + the two-stage output, sigmoid focal
classification loss, L1 + GIoU box losses on the matched pairs and on the denoising queries, KL distillation against the teacher's
box logits), so that ONE number covers a whole forward + backward of the path:

    ResNet-50 (layer2-4 trained) -> input projections -> 6 encoder layers -> two-stage class score + top-900 -> denoising layout ->
    6 decoder layers (+ box refinement) -> class / box / distillation heads -> frozen teacher -> ROIAlign -> attention pool ->
    matcher -> losses -> backward through all of it.

bf16 activations wherever the library has a bf16 path (backbone, encoder, decoder), fp32 master parameters.  The optimizer step and
the data pipeline are not part of it (and not part of this repository's scope).
"""
import math
import sys
import time

import torch
import torch.nn.functional as F
from torch import nn

from richsem_amd import workload as W
from richsem_amd.backbone import InputProjection, ResNet50
from richsem_amd.capture import quiet_gc
from richsem_amd.clip_resnet import ModifiedResNetTeacher
from richsem_amd.dn import prepare_dn_layout
from richsem_amd.functions.linear import Lin256Function, VersionCache, pack_linear256
from richsem_amd.matcher import BoxPairLoss, FocalNegativeSum, FocalPositiveSum, HungarianMatcher
from richsem_amd.modules import (MLP, refine_boxes, DeformableTransformerDecoderLayer, DeformableTransformerEncoderLayer, TransformerDecoder,
                                 clip_box_targets, get_reference_points, inverse_sigmoid)
from richsem_amd.two_stage import ClassScorer

NUM_CLASSES, NUM_QUERIES, DN_NUMBER, PROJ = 1204, 900, 100, 1024


# ---- small helpers of the reference, restated (host-side torch ops) -----------------------------------------------------------------
def box_cxcywh_to_xyxy(x):
    cx, cy, w, h = x.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], -1)


def giou_pairs(a, b):
    """generalised IoU of matched pairs (util/box_ops.py:41-64 on the diagonal), xyxy boxes"""
    area = lambda t: (t[:, 2] - t[:, 0]) * (t[:, 3] - t[:, 1])
    lt, rb = torch.max(a[:, :2], b[:, :2]), torch.min(a[:, 2:], b[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, 0] * wh[:, 1]                 # (not .prod(-1): its backward reads a zero count back to the host)
    union = area(a) + area(b) - inter
    iou = inter / (union + 1e-6)
    hw = (torch.max(a[:, 2:], b[:, 2:]) - torch.min(a[:, :2], b[:, :2])).clamp(min=0)
    hull = hw[:, 0] * hw[:, 1]
    return iou - (hull - union) / (hull + 1e-6)


def sine_position(mask, num_pos_feats=128, temperature=20.0):
    """PositionEmbeddingSineHW (models/richsem/position_encoding.py:46-92, temperatureH = temperatureW = 20, normalize): (N, H, W) bool
    padding mask -> (N, H * W, 256)"""
    not_mask = ~mask
    y_embed, x_embed = not_mask.cumsum(1, dtype=torch.float32), not_mask.cumsum(2, dtype=torch.float32)
    eps, scale = 1e-6, 2 * math.pi
    y_embed = y_embed / (y_embed[:, -1:, :] + eps) * scale
    x_embed = x_embed / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32, device=mask.device)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / num_pos_feats)
    px, py = x_embed[:, :, :, None] / dim_t, y_embed[:, :, :, None] / dim_t
    px = torch.stack((px[:, :, :, 0::2].sin(), px[:, :, :, 1::2].cos()), dim=4).flatten(3)
    py = torch.stack((py[:, :, :, 0::2].sin(), py[:, :, :, 1::2].cos()), dim=4).flatten(3)
    return torch.cat((py, px), dim=3).flatten(1, 2)


def encoder_output_proposals(memory_padding_mask, shapes):
    """models/richsem/utils.py:10-65, the part that depends on the batch's geometry only: per-pixel anchor boxes (unsigmoided, +inf at
    padded / out-of-range positions) and the (N, S, 1) mask of the positions whose memory is zeroed"""
    N, dev = memory_padding_mask.shape[0], memory_padding_mask.device
    proposals, cur = [], 0
    for lvl, (H_, W_) in enumerate(shapes):
        m = memory_padding_mask[:, cur:cur + H_ * W_].view(N, H_, W_, 1)
        valid_H, valid_W = (~m[:, :, 0, 0]).sum(1), (~m[:, 0, :, 0]).sum(1)
        gy, gx = torch.meshgrid(torch.linspace(0, H_ - 1, H_, dtype=torch.float32, device=dev),
                                torch.linspace(0, W_ - 1, W_, dtype=torch.float32, device=dev), indexing="ij")
        grid = torch.cat([gx.unsqueeze(-1), gy.unsqueeze(-1)], -1)
        scale = torch.cat([valid_W.unsqueeze(-1), valid_H.unsqueeze(-1)], 1).view(N, 1, 1, 2)
        grid = (grid.unsqueeze(0).expand(N, -1, -1, -1) + 0.5) / scale
        wh = torch.ones_like(grid) * 0.05 * (2.0 ** lvl)
        proposals.append(torch.cat((grid, wh), -1).view(N, -1, 4))
        cur += H_ * W_
    out = torch.cat(proposals, 1)
    valid = ((out > 0.01) & (out < 0.99)).all(-1, keepdim=True)
    out = torch.log(out / (1 - out))
    zeroed = memory_padding_mask.unsqueeze(-1) | ~valid
    return out.masked_fill(zeroed, float("inf")), zeroed


class Step(nn.Module):
    """the rows with their (synthetic) parameters; ``forward`` = model forward + criterion, returns the loss and section times"""

    def __init__(self, n_img=2, height=800, width=1333, boxes_per_image=12, seed=0, dev="cuda"):
        super().__init__()
        torch.manual_seed(seed)
        self.n_img, self.H, self.Wimg, self.K = n_img, height, width, boxes_per_image
        self.Wpad = (width + 31) // 32 * 32
        self.backbone = ResNet50()
        self.backbone.load_state_dict(W.resnet50_state_dict(seed=seed + 1))
        self.input_proj = InputProjection()
        enc = DeformableTransformerEncoderLayer(256, 2048, dropout=0.0, n_levels=4, n_heads=8, n_points=4)
        self.encoder = nn.ModuleList([enc] + [type(enc)(256, 2048, dropout=0.0, n_levels=4, n_heads=8, n_points=4) for _ in range(5)])
        self.level_embed = nn.Parameter(torch.randn(4, 256))
        self.enc_output, self.enc_output_norm = nn.Linear(256, 256), nn.LayerNorm(256)
        self.enc_out_bbox_embed = MLP(256, 256, 4, 3)
        self.tgt_embed = nn.Embedding(NUM_QUERIES, 256)
        self.label_enc = nn.Embedding(NUM_CLASSES + 1, 256)
        dec = DeformableTransformerDecoderLayer(256, 2048, dropout=0.0, n_levels=4, n_heads=8, n_points=4)
        self.decoder = TransformerDecoder(dec, 6, nn.LayerNorm(256), d_model=256)
        self.decoder.bbox_embed = nn.ModuleList([MLP(256, 256, 4, 3) for _ in range(6)])
        # the CLIP-space classifier (richsem.py:38-205): visual projection (trained), distillation projection (trained), frozen text side
        self.dino_visual_proj = nn.Linear(256, PROJ, bias=False)
        self.proj_dino_hs = nn.Linear(256, PROJ)
        self.register_buffer("text_embed", torch.randn(NUM_CLASSES, PROJ))
        self.register_buffer("logit_scale", torch.tensor(math.log(1 / 0.07)))
        for m in list(self.encoder) + list(self.decoder.layers):
            att = m.self_attn if hasattr(m.self_attn, "sampling_offsets") else m.cross_attn
            with torch.no_grad():
                att.sampling_offsets.weight.normal_(0, 0.01)
                att.attention_weights.weight.normal_(0, 0.05)
        self.to(dev)
        self.teacher = ModifiedResNetTeacher(W.clip_rn50_state_dict(seed=seed + 2), heads=32)      # frozen: not a sub-module
        self.matcher = HungarianMatcher(cost_class=2.0, cost_bbox=5.0, cost_giou=2.0)
        self.scorer = ClassScorer(2)
        self.times = {}
        self._eo_pack = VersionCache()
        self.timing = True
        self.stop_at = None       # (tools/capture_probe.py: end the step after this section with a surrogate loss)
        # bfloat16: the library's bf16 rows (the step bench.py times).  float32: the SAME parameters through the reference's op sequence in
        # fp32 from the input projections on (PyTorch ops around the operator's fp32 entry points; the backbone stays on the bf16
        # convolution kernels, its output is cast) -- the yardstick of tests/test_gpu_step.py
        self.act_dtype = torch.bfloat16

    # synthetic LVIS-shaped batch
    def batch(self, seed=0):
        g = torch.Generator().manual_seed(42 + seed)
        N, dev = self.n_img, self.level_embed.device
        images = torch.zeros(N, 3, self.H, self.Wpad)
        images[..., :self.Wimg] = torch.randn(N, 3, self.H, self.Wimg, generator=g)
        mask = torch.zeros(N, self.H, self.Wpad, dtype=torch.bool)
        mask[..., self.Wimg:] = True
        targets = []
        for _ in range(N):
            cxcy, wh = torch.rand(self.K, 2, generator=g) * 0.6 + 0.2, torch.rand(self.K, 2, generator=g) * 0.35 + 0.05
            targets.append({"boxes": torch.cat((cxcy, wh), 1).to(dev), "labels": torch.randint(1, 1204, (self.K,), generator=g).to(dev),
                            "size": torch.tensor([float(self.H), float(self.Wimg)], device=dev)})
        return images.to(dev), mask.to(dev), targets

    def _mark(self, name):
        if not self.timing:       # (timing events cannot be recorded while the stream is being captured)
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self._events.append((name, ev))

    def class_logits(self, hs):
        """CLIPAlign.forward / forward_hs (richsem.py:182-205)"""
        f = F.linear(hs, self.dino_visual_proj.weight.to(hs.dtype))
        f = f / f.norm(dim=-1, keepdim=True)
        t = self.text_embed / self.text_embed.norm(dim=-1, keepdim=True)
        return (self.logit_scale.exp() * (f @ t.to(f.dtype).t())).float()

    @torch.no_grad()
    def prepare(self, mask, targets):
        """what depends on the batch's geometry only (level shapes, padding masks, valid ratios, encoder reference points, denoising
        layout) and on the frozen text side (the two-stage scorer's packed operand): built once per batch shape -- a few small
        launches and host -> device copies a trainer repeats per step; kept out of the step so that it can be captured into a graph"""
        dev = mask.device
        shapes = list(W.pyramid_shapes(self.H, self.Wpad))
        masks = [F.interpolate(mask[None].float(), size=s).to(torch.bool)[0] for s in shapes]
        spatial = torch.tensor(shapes, dtype=torch.int64, device=dev)
        st = {"shapes": shapes, "masks": masks, "spatial": spatial,
              "lsi": torch.cat((spatial.new_zeros(1), spatial.prod(1).cumsum(0)[:-1])),
              "mask_flat": torch.cat([m.flatten(1) for m in masks], 1),
              "valid_ratios": torch.stack([torch.stack([(~m[:, 0, :]).sum(1) / m.shape[2], (~m[:, :, 0]).sum(1) / m.shape[1]], -1)
                                           for m in masks], 1).float()}
        st["ref"] = get_reference_points(shapes, st["valid_ratios"], dev)
        st["pos_sine"] = torch.cat([sine_position(m) for m in masks], 1)                               # (N, S, 256): the masks' part of pos
        level_of = torch.cat([torch.full((h * w,), l, dtype=torch.int64, device=dev) for l, (h, w) in enumerate(shapes)])
        st["level_onehot"] = F.one_hot(level_of, len(shapes)).float()      # (S, L): the level embedding as a product (its backward a 4-row GEMM, not 22 k serialised row adds)
        st["proposals"], st["zeroed"] = encoder_output_proposals(st["mask_flat"], shapes)
        st["known_num"] = [len(t["labels"]) for t in targets]
        st["lay"] = prepare_dn_layout(st["known_num"], DN_NUMBER, NUM_QUERIES, use_cdn=True)
        st["scale"] = self.logit_scale.detach().clone()
        self.scorer.prepare(self.dino_visual_proj.weight, self.text_embed, self.logit_scale)      # (once per weight update)
        self.static = st
        self._targets = targets
        return st

    def forward(self, images, mask, targets, indices=None, topk=None):
        """``indices``: the matcher's assignment held fixed (graph capture, A/B tests); ``topk``: the two-stage selection held fixed"""
        adt = self.act_dtype
        self._events = []
        self._mark("start")
        N, dev = images.shape[0], images.device
        st = self.static
        shapes, masks, spatial, lsi, mask_flat, valid_ratios = (st[k] for k in ("shapes", "masks", "spatial", "lsi", "mask_flat", "valid_ratios"))
        # ---- backbone + input projections (richsem.py:581-612) -------------------------------------------------------------------
        feats = self.backbone(images)
        self._mark("backbone")
        if self.stop_at == "backbone":
            return sum(f.float().sum() for f in feats)
        srcs, got_shapes = self.input_proj(feats, out_dtype=adt)
        assert got_shapes == shapes
        src = torch.cat(srcs, 1)
        pos_flat = (st["pos_sine"] + st["level_onehot"] @ self.level_embed).to(adt)           # sine part + level embedding (:596-612)
        self._mark("input_proj")
        if self.stop_at == "input_proj":
            return src.float().sum() + pos_flat.float().sum()
        # ---- encoder (deformable_transformer.py:319) -------------------------------------------------------------------------------
        ref = st["ref"]
        memory = src
        for layer in self.encoder:
            memory = layer(memory, pos_flat, ref, spatial, lsi, mask_flat)
        self._mark("encoder")
        if self.stop_at == "encoder":
            return memory.float().sum()
        # ---- two-stage query selection (:352-380) ----------------------------------------------------------------------------------
        eo = self.enc_output
        if adt == torch.bfloat16:
            pk = self._eo_pack.get((eo.weight, eo.bias), lambda: pack_linear256([eo.weight], [eo.bias]))
            output_memory = Lin256Function.apply(memory.masked_fill(st["zeroed"], 0.0), pk, None, False, eo.weight, eo.bias)   # bf16, lin256
        else:
            output_memory = eo(memory.masked_fill(st["zeroed"], 0.0))
        output_memory = F.layer_norm(output_memory, (256,), self.enc_output_norm.weight.to(adt), self.enc_output_norm.bias.to(adt),
                                     self.enc_output_norm.eps)
        if topk is None:
            topk = self.scorer.topk_proposals(output_memory, NUM_QUERIES)                              # no logit tensor (two_stage.py)
        self.last_topk = topk
        coord_unselected = self.enc_out_bbox_embed(output_memory).float() + st["proposals"]
        refpoint_undetach = torch.gather(coord_unselected, 1, topk[..., None].expand(-1, -1, 4))
        tgt_undetach = torch.gather(output_memory, 1, topk[..., None].expand(-1, -1, 256)).float()
        interm = {"pred_logits": self.class_logits(tgt_undetach), "pred_boxes": refpoint_undetach.sigmoid()}
        self._mark("two_stage")
        if self.stop_at == "two_stage":
            return interm['pred_logits'].sum() + interm['pred_boxes'].sum()
        # ---- denoising queries (dn_components.py:11-193): layout on the library's kernels, noise as torch ops -----------------------
        known_num, lay = st["known_num"], st["lay"]
        pad, groups = lay["pad_size"], lay["num_dn_group"]
        labels, boxes = torch.cat([t["labels"] for t in targets]), torch.cat([t["boxes"] for t in targets])
        known_labels, known_boxes = labels.repeat(2 * groups), boxes.repeat(2 * groups, 1)
        fz = self.frozen_noise      # (tests: the same draws in every step and in every form of the step -- see freeze_noise)
        p = torch.rand(known_labels.shape, device=dev) if fz is None else fz["p"]
        rnd_lab = torch.randint(0, NUM_CLASSES, known_labels.shape, device=dev) if fz is None else fz["labels"]
        noised = torch.where(p < 0.25, rnd_lab, known_labels)
        xyxy = box_cxcywh_to_xyxy(known_boxes)
        diff = torch.cat((known_boxes[:, 2:] / 2, known_boxes[:, 2:] / 2), 1)
        sign = (torch.randint(0, 2, xyxy.shape, device=dev).float() * 2 - 1) if fz is None else fz["sign"]
        rand_part = torch.rand(xyxy.shape, device=dev) if fz is None else fz["rand"]
        neg = (torch.arange(known_labels.numel(), device=dev) // labels.numel()) % 2 == 1
        rand_part = torch.where(neg[:, None], rand_part + 1.0, rand_part) * sign
        xyxy = (xyxy + rand_part * diff).clamp(0.0, 1.0)
        nb = torch.cat(((xyxy[:, :2] + xyxy[:, 2:]) / 2, xyxy[:, 2:] - xyxy[:, :2]), 1)
        q_label = torch.zeros(N, pad, 256, device=dev)
        q_bbox = torch.zeros(N, pad, 4, device=dev)
        q_label[lay["known_bid"], lay["map_known_indice"]] = self.label_enc(noised)
        q_bbox[lay["known_bid"], lay["map_known_indice"]] = inverse_sigmoid(nb)
        tgt = torch.cat((q_label, self.tgt_embed.weight[None].expand(N, -1, -1)), 1)                   # embed_init_tgt
        refpoints = torch.cat((q_bbox, refpoint_undetach.detach()), 1)
        self._mark("dn")
        if self.stop_at == "dn":
            return tgt.sum() + refpoints.sum() + interm['pred_logits'].sum() + interm['pred_boxes'].sum()
        # ---- decoder (:427) -----------------------------------------------------------------------------------------------------------
        hs, refs = self.decoder(tgt=tgt.transpose(0, 1).to(adt), memory=memory.transpose(0, 1), tgt_mask=lay["attn_mask"],
                                memory_key_padding_mask=mask_flat, refpoints_unsigmoid=refpoints.transpose(0, 1), level_start_index=lsi,
                                spatial_shapes=spatial, valid_ratios=valid_ratios)
        self._mark("decoder")
        if self.stop_at == "decoder":
            return sum(h.float().sum() for h in hs) + sum(r.sum() for r in refs) + interm['pred_logits'].sum() + interm['pred_boxes'].sum()
        # ---- heads (richsem.py:705-733) -------------------------------------------------------------------------------------------------
        hs_stack = torch.stack(hs)                                                                     # (6, N, pad + 900, 256)
        coords = torch.stack([refine_boxes(self.decoder.bbox_embed[l](hs[l]), refs[l]) for l in range(6)])      # richsem.py:705-715
        logits = self.class_logits(hs_stack)                                                           # (6, N, 1092, 1204)
        clip_hs = F.linear(hs[-1], self.proj_dino_hs.weight.to(hs[-1].dtype), self.proj_dino_hs.bias.to(hs[-1].dtype))
        clip_hs = clip_hs / clip_hs.norm(dim=-1, keepdim=True)
        tn = self.text_embed / self.text_embed.norm(dim=-1, keepdim=True)
        clip_logits = (self.logit_scale.exp() * (clip_hs @ tn.to(clip_hs.dtype).t())).float()          # (N, 1092, 1204)
        self._mark("heads")
        if self.stop_at == "heads":
            return coords.sum() + logits.sum() + clip_logits.sum() + interm['pred_logits'].sum() + interm['pred_boxes'].sum()
        # ---- frozen teacher -> ROIAlign -> attention pool -> text logits (richsem.py:614-629, :741-768) ----------------------------------
        if self._model_only == "student":      # (model_part without the frozen teacher: run_graphed launches it beside the host's matching)
            return logits, coords, interm["pred_logits"], interm["pred_boxes"], clip_logits
        t_logits = self.teacher_part(images, targets)
        self._mark("teacher")
        if self._model_only:      # (model_part: the step up to the matcher, as tensors)
            return logits, coords, interm["pred_logits"], interm["pred_boxes"], clip_logits, t_logits
        # ---- matcher (matcher.py:30-78, one host copy for the 7 outputs) -------------------------------------------------------------------
        if indices is None:
            indices = self.match(logits, coords, interm["pred_logits"], interm["pred_boxes"], targets)
        packed = indices if (isinstance(indices, tuple) and torch.is_tensor(indices[0])) else self.pack_indices(indices, targets)
        if packed is not indices:
            self.last_indices = indices
        self._mark("matcher")
        loss = self.loss_part(logits, coords, interm["pred_logits"], interm["pred_boxes"], clip_logits, t_logits, *packed)
        self._mark("criterion")
        return loss

    _model_only = False
    frozen_noise = None

    def freeze_noise(self, seed):
        """draw the denoising noise ONCE from ``seed`` and use it in every later step (a training step draws it anew each time,
        dn_components.py:72-110): eager, captured and graphed forms of the step can then be compared parameter by parameter"""
        st = self.static
        n = 2 * st["lay"]["num_dn_group"] * sum(st["known_num"])
        g = torch.Generator(device=self.level_embed.device).manual_seed(seed)
        dev = self.level_embed.device
        self.frozen_noise = {"p": torch.rand(n, device=dev, generator=g), "labels": torch.randint(0, NUM_CLASSES, (n,), device=dev, generator=g),
                             "sign": torch.randint(0, 2, (n, 4), device=dev, generator=g).float() * 2 - 1,
                             "rand": torch.rand((n, 4), device=dev, generator=g)}

    def model_part(self, images, mask=None, targets=None, teacher=True):
        """the step up to the matcher: -> (logits (6, N, Q, C), boxes (6, N, Q, 4), two-stage logits, two-stage boxes, distillation logits,
        the teacher's box logits -- unless ``teacher`` is False): tensors in, tensors out, nothing read back to the host -- the part
        ``run_graphed`` captures"""
        self._model_only = True if teacher else "student"
        try:
            return self.forward(images, mask, self._targets if targets is None else targets)
        finally:
            self._model_only = False

    @torch.no_grad()
    def teacher_part(self, images, targets=None):
        """frozen CLIP-RN50 teacher -> ROIAlign -> attention pool -> text logits of the target boxes (richsem.py:614-629, :741-768): no
        gradient, no dependence on the student"""
        targets = self._targets if targets is None else targets
        _, fmap = self.teacher(images, ret_sp=True)
        _, t_logits = clip_box_targets(fmap, targets, self.teacher.attnpool, self.text_embed, self.static["scale"])
        return torch.cat(t_logits).float()

    def _match_outputs(self, logits, coords, il, ib):
        pad = self.static["lay"]["pad_size"]
        return [{"pred_logits": logits[l][:, pad:], "pred_boxes": coords[l][:, pad:]} for l in range(logits.shape[0])] + \
            [{"pred_logits": il, "pred_boxes": ib}]

    def match(self, logits, coords, il, ib, targets):
        return self.matcher.match_many(self._match_outputs(logits, coords, il, ib), targets)

    def match_begin(self, logits, coords, il, ib, targets):
        """the device half of :meth:`match` (cost blocks + copy to the host enqueued, nothing waits); :meth:`match_end` the host half"""
        return self.matcher.match_many_begin(self._match_outputs(logits, coords, il, ib), targets)

    def match_end(self, pending):
        return self.matcher.match_many_end(pending)

    def pack_indices(self, indices, targets):
        """the Hungarian assignment of the 6 + 1 outputs as flat device tensors of a size the batch fixes (every target is matched once per
        output): (labels, boxes) of all targets, (layer, image, query, target) of the decoder outputs' pairs, (image, query, target) of the
        two-stage output's and of the last layer's (the distillation pairs)"""
        dev = self.level_embed.device
        off = [0]
        for t in targets:
            off.append(off[-1] + len(t["labels"]))
        labels, boxes = torch.cat([t["labels"] for t in targets]), torch.cat([t["boxes"] for t in targets])

        def flat(idx_list):
            li = torch.cat([torch.full((len(s),), k, dtype=torch.int64) for k, idx in enumerate(idx_list) for s, _ in idx])
            bi = torch.cat([torch.full((len(s),), b, dtype=torch.int64) for idx in idx_list for b, (s, _) in enumerate(idx)])
            si = torch.cat([s.cpu() for idx in idx_list for s, _ in idx])
            tj = torch.cat([j.cpu() + off[b] for idx in idx_list for b, (_, j) in enumerate(idx)])
            return torch.stack((li, bi, si, tj)).to(dev, non_blocking=True)
        nl = len(indices) - 1
        return labels, boxes, flat(indices[:nl]), flat(indices[nl:]), flat(indices[nl - 1:nl])

    def loss_part(self, logits, coords, il, ib, clip_logits, t_logits, labels, boxes, m_dec, m_int, m_dis):
        """criterion (richsem.py:1124-1306, compact): the same per-output sums as the reference's loop over the 6 + 1 outputs -- sigmoid focal
        loss (over every query incl. the denoising part's negative slots, whose target is no-object; the federated-loss class sampling of
        use_fed_loss is not restated: all classes count), L1 + GIoU on the matched pairs and on the denoising queries' positive slots, KL
        distillation -- formed in ONE pass per kind over
        the stacked outputs: the all-negative focal term of a whole logit tensor is one kernel each way (matcher.FocalNegativeSum), and the
        positive entries / box pairs of the matched, two-stage and denoising parts are concatenated with a weight each (1 / num_boxes, or
        1 / (num_boxes x groups)) so that every loss formula runs once -- as one kernel each (matcher.FocalPositiveSum, matcher.BoxPairLoss:
        value and gradient in one launch; a loop over the outputs is ~40 small launches per output and kind).
        Tensors in, the loss out."""
        st = self.static
        dev = logits.device
        lay = st["lay"]
        pad, groups, single = lay["pad_size"], lay["num_dn_group"], lay["single_pad"]
        num_boxes = float(max(sum(st["known_num"]), 1))
        nbx = num_boxes * groups
        nl, N, Q = logits.shape[0], logits.shape[1], logits.shape[2]
        alpha = 0.25
        cst = st.get("loss_static")
        if cst is None or cst["key"] != (nl, N, Q, pad, groups, single):      # index / weight tensors the batch's geometry fixes
            pos_slots = (torch.arange(groups, device=dev)[:, None] * 2 * single + torch.arange(single, device=dev)[None]).flatten()
            # (reference richsem.py:938-964: loss_labels of the denoising part runs the focal loss over ALL pad_size denoising queries --
            # the negative slots with the no-object target -- normalised by num_boxes x groups, :1180 / :1227; the matching part by num_boxes)
            w_q = torch.empty(Q, dtype=torch.float32, device=dev)
            w_q[:pad] = 1.0 / nbx
            w_q[pad:] = 1.0 / num_boxes
            n_dn = nl * N * pos_slots.numel()
            cst = {"key": (nl, N, Q, pad, groups, single), "pos_slots": pos_slots,
                   "w_rows": w_q[None, None, :].expand(nl, N, Q).contiguous(),
                   "w_int": torch.full(il.shape[:2], 1.0 / num_boxes, dtype=torch.float32, device=dev),
                   "dn_l": torch.arange(nl, device=dev)[:, None, None].expand(nl, N, pos_slots.numel()).reshape(-1),
                   "dn_n": torch.arange(N, device=dev)[None, :, None].expand(nl, N, pos_slots.numel()).reshape(-1),
                   "dn_q": pos_slots[None, None, :].expand(nl, N, -1).reshape(-1),
                   "w_dn": torch.full((n_dn,), 1.0 / nbx, dtype=torch.float32, device=dev)}
            st["loss_static"] = cst
        # ---- classification: all-negative term of every entry, then what the positive entries contribute instead ------------------------
        loss = FocalNegativeSum.apply(logits, cst["w_rows"], alpha) + FocalNegativeSum.apply(il, cst["w_int"], alpha)
        li, bi, si, tj = m_dec
        _, ibi, isi, itj = m_int
        dn_lab = labels.view(N, -1).repeat(1, groups)[None].expand(nl, -1, -1).reshape(-1)      # every image has `single` boxes here
        x_pos = torch.cat((logits[li, bi, si + pad, labels[tj]], il[ibi, isi, labels[itj]], logits[cst["dn_l"], cst["dn_n"], cst["dn_q"], dn_lab]))
        w_pair = torch.cat((torch.full((li.numel() + ibi.numel(),), 1.0 / num_boxes, dtype=torch.float32, device=dev), cst["w_dn"]))
        loss = loss + FocalPositiveSum.apply(x_pos, w_pair, alpha)
        # ---- boxes: L1 + GIoU of all pairs at once -------------------------------------------------------------------------------------------
        pb = torch.cat((coords[li, bi, si + pad], ib[ibi, isi], coords[cst["dn_l"], cst["dn_n"], cst["dn_q"]]))
        tb = torch.cat((boxes[tj], boxes[itj], boxes.view(N, -1, 4).repeat(1, groups, 1)[None].expand(nl, -1, -1, -1).reshape(-1, 4)))
        loss = loss + BoxPairLoss.apply(pb, tb, w_pair, 5.0, 2.0)
        # ---- distillation: KL of the matched queries' CLIP logits against the teacher's box logits (richsem.py:1255-1300) -------------------
        _, bi, si, tj = m_dis
        loss = loss + 0.5 * F.kl_div(F.log_softmax(clip_logits[bi, si + pad], -1), F.softmax(t_logits[tj], -1), reduction="batchmean")
        return loss

    def section_ms(self):
        """ms per forward section of the last step (events recorded on the current stream)"""
        torch.cuda.synchronize()
        out, prev = {}, self._events[0][1]
        for name, ev in self._events[1:]:
            out[name] = prev.elapsed_time(ev)
            prev = ev
        return out


def pin_grad_accumulators(params):
    """Create every parameter's AccumulateGrad node NOW -- on the current stream -- and return the nodes; the caller keeps them alive.
    The autograd engine runs an AccumulateGrad node on the stream that was current when the node was CREATED, and a node lives as long as
    some graph (or this list) references it.  Round 4's harness let the first forward that happened to touch a parameter decide -- for
    torch.cuda.make_graphed_callables that is its private warm-up stream -- and a later backward on the capture stream then synchronised
    with that foreign stream on every step ("AccumulateGrad node's stream does not match ..."), the precondition of the
    hipStreamEndCapture crash of profiles/r04_capture_probe.txt.  Pinned here, every later graph reuses these nodes."""
    return [p.view_as(p).grad_fn.next_functions[0][0] for p in params if p.requires_grad]


CLIP_MAX_NORM = 0.1      # reference config/RichSem/baseline_4scale.py:19 (clip_max_norm), engine.py:110-112


def make_optimizer(params):
    """AdamW as the reference builds it (main.py:213-214; lr / weight_decay of config/RichSem/baseline_4scale.py:7,14), one fused launch per step"""
    return torch.optim.AdamW(params, lr=1e-4, weight_decay=1e-4, fused=True)


def optimizer_step(opt, params):
    """what follows ``losses.backward()`` in the reference's training loop (engine.py:109-113): gradient clipping to clip_max_norm, AdamW step.
    No host synchronisation: the norm stays on the device (``error_if_nonfinite`` is off as in the reference's call)."""
    torch.nn.utils.clip_grad_norm_(params, CLIP_MAX_NORM, foreach=True)
    opt.step()


def run(n_img, dev, steps=5, warmup=2, graph=True, stop_at=None):
    """time `steps` composed steps (forward + loss + backward); returns the dict bench.py attaches as ``full_step``"""
    model = Step(n_img=n_img, dev=dev)
    model.stop_at = stop_at      # (profiling aid: the step cut off after a section, see tools/step_sections.sh)
    images, mask, targets = model.batch()
    model.prepare(mask, targets)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = make_optimizer(params)

    def step(indices=None, optimize=True):
        for p in params:
            p.grad = None
        loss = model(images, mask, targets, indices)
        fwd = None
        if model.timing:
            fwd = torch.cuda.Event(enable_timing=True)
            fwd.record()
        loss.backward()
        if optimize:
            optimizer_step(opt, params)
        return loss, fwd

    # Everything -- warm-up, the timed eager steps, the capture -- runs on ONE side stream (round 4).  An autograd graph pins every
    # parameter's AccumulateGrad node to the stream it was built on; round 3 built the eager steps on the default stream and captured on
    # another one with the last step's `loss` still referenced: the engine then synchronises the capture stream with the foreign stream
    # inside the capture ("AccumulateGrad node's stream does not match ..."), and this ROCm build segfaults in hipStreamEndCapture instead
    # of failing the capture (tools/capture_crash_probe.py, profiles/r04_capture_probe.txt: only the variants that keep `loss` alive crash,
    # and none does when the eager phase already ran on the capture stream).  The library's workspaces are per (device, stream) as well:
    # a capture on a cold stream records the fallback kernels (round 3's 31.0 ms against 27.1 ms on the warmed stream).
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    rows, totals, bwds = {}, [], []
    with torch.cuda.stream(side):
        pinned = pin_grad_accumulators(params)      # (kept alive to the end of run(): every graph below reuses these nodes)
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        for _ in range(steps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            loss, fwd = step()
            b.record()
            torch.cuda.synchronize()
            totals.append(a.elapsed_time(b))
            bwds.append(fwd.elapsed_time(b))
            for k, v in model.section_ms().items():
                rows.setdefault(k, []).append(v)
    torch.cuda.current_stream().wait_stream(side)
    ms = sum(totals) / len(totals)
    out = {"what": "ONE composed training step on the library's rows at configs[1] sizes: ResNet-50 (layer2-4 trained) -> input projections "
                   "-> 6 encoder layers -> two-stage score + top-900 -> denoising layout -> 6 decoder layers -> heads -> frozen CLIP-RN50 "
                   "teacher -> ROIAlign -> attention pool -> matcher -> losses, forward + backward, gradient clipping (0.1) + fused AdamW step "
                   "(reference engine.py:105-113), bf16 activations / fp32 parameters; synthetic weights and batch; no data pipeline "
                   "(bench_step.py)",
           "ms": round(ms, 2), "img_per_s": round(n_img / (ms * 1e-3), 2), "loss": float(loss.detach()),
           "forward_rows_ms": {k: round(sum(v) / len(v), 3) for k, v in rows.items()},
           "backward_ms": round(sum(bwds) / len(bwds), 2)}
    if not graph:
        return out
    # the device part as a captured graph: the assignment of the last eager step held fixed (the matcher's host round trip cannot be captured)
    indices = model.pack_indices(model.last_indices, targets)      # (device tensors: no host <-> device copies in the capture)
    torch.cuda.synchronize()
    model.timing = False

    def replay_ms(stop_at):
        """capture the step cut off after section `stop_at` (None: all of it) into a HIP graph; ms per replay"""
        model.stop_at = stop_at
        with torch.cuda.stream(side):
            for _ in range(2):
                step(indices, optimize=False)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with quiet_gc(), torch.cuda.graph(g, stream=side):      # the stream that was warmed up
            step(indices, optimize=False)
        g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    try:
        rep = replay_ms(None)
        out["graph_replay"] = {"ms": round(rep, 2), "img_per_s": round(n_img / (rep * 1e-3), 2),
                               "note": "forward + backward (no optimizer step: a diagnostic of the device work per row) captured once into a HIP "
                                       "graph and replayed, the Hungarian assignment of the last eager step held fixed (its host round trip "
                                       "cannot be captured)"}
        # GPU time per row, forward + backward: replay time of the step cut off after each section (a surrogate loss = the sum of the
        # section's outputs), differenced; "criterion" = the rest (teacher + matcher-less losses and their backward into the heads)
        prev, table = 0.0, {}
        for name in ("backbone", "input_proj", "encoder", "two_stage", "dn", "decoder", "heads"):
            t = replay_ms(name)
            table[name] = round(t - prev, 2)
            prev = t
        table["teacher+criterion"] = round(rep - prev, 2)
        out["graph_replay"]["rows_fwd_bwd_ms"] = table
    except Exception as e:      # a capture failure must not take the bench line down
        import traceback
        traceback.print_exc(file=sys.stderr)
        out.setdefault("graph_replay", {})["error"] = f"{type(e).__name__}: {str(e)[:300]}"
    finally:
        model.stop_at = None
    try:
        del model
        torch.cuda.empty_cache()
        out["graphed_sections"] = run_graphed(n_img, dev, steps=steps, warmup=warmup)
    except Exception as e:      # noqa: BLE001
        import traceback
        traceback.print_exc(file=sys.stderr)
        out["graphed_sections"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
    return out


class _ModelPart(nn.Module):
    """the step up to the matcher as a module of tensors (what torch.cuda.make_graphed_callables captures, forward and backward)"""

    def __init__(self, step):
        super().__init__()
        self.step = step

    def forward(self, images):
        return self.step.model_part(images, self.step._mask, teacher=False)


class _LossPart(nn.Module):
    def __init__(self, step):
        super().__init__()
        self.step = [step]      # (not a sub-module: the criterion has no parameters of its own)

    def forward(self, *tensors):
        return self.step[0].loss_part(*tensors)


def run_graphed(n_img, dev, steps=5, warmup=2, optimizer=True, noise_seed=None, return_grads=False, **step_kwargs):
    """The composed step as a trainer can run it WITHOUT freezing the matcher: the two device-only parts -- everything up to the matcher,
    and the criterion -- each captured once, forward and backward, with ``torch.cuda.make_graphed_callables`` (HIP graphs replayed by
    autograd), the Hungarian assignment between them live on the host every step.  Eagerly the step is bound by ~2900 kernel launches
    (ms above); this is the same work with three launches' worth of host time.  Returns the dict bench.py attaches as
    ``full_step.graphed_sections`` (``step_kwargs``: a smaller Step for the tests)."""
    model = Step(n_img=n_img, dev=dev, **step_kwargs)
    model.timing = False
    images, mask, targets = model.batch()
    model.prepare(mask, targets)
    model._mask = mask
    if noise_seed is not None:
        model.freeze_noise(noise_seed)
    part_a, part_b = _ModelPart(model), _LossPart(model)
    # ONE side stream for the eager warm-up, the captures and the training steps: torch captures on its class-wide capture stream, which is
    # set to that stream here -- the library's workspaces are per (device, stream) and are not allocated during capture (on a cold stream
    # the capture records the fallback kernels), and an autograd graph built on another stream pins the parameters' AccumulateGrad nodes
    # there (the capture then synchronises with that stream from inside: wrong results at best, see run())
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    saved_capture_stream = torch.cuda.graph.default_capture_stream
    torch.cuda.graph.default_capture_stream = side
    try:
        with torch.cuda.stream(side):
            pinned = pin_grad_accumulators(model.parameters())      # on `side`, before anything touches a parameter; alive to the end
            with torch.no_grad():
                outs = model.model_part(images, mask)
            idx = model.pack_indices(model.match(*outs[:4], targets), targets)
            model.loss_part(*model.model_part(images, mask), *idx).backward()      # (eager once: workspaces of this stream, caches)
            for p in model.parameters():
                p.grad = None
            sample_b = tuple(o.detach().clone().requires_grad_(i < 5) for i, o in enumerate(outs)) + tuple(idx)
            torch.cuda.synchronize()
            # (make_graphed_callables warms its callables up on a PRIVATE stream of its own making -- torch/cuda/graphs.py -- before it
            # captures them on `side`: during those three iterations the gradients arrive at the pinned nodes from that other stream, by
            # construction.  The engine's warning about it is switched off for this one call only; every training step below runs with it
            # on, and tests/test_gpu_step.py fails on it)
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
            try:
                with quiet_gc():      # (richsem_amd/capture.py: a collection inside a capture aborts the process)
                    ga, gb = torch.cuda.make_graphed_callables((part_a, part_b), ((images,), sample_b), num_warmup_iters=3, allow_unused_input=True)
            finally:
                torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(True)
            # the frozen teacher (no gradient, independent of the student) is a HIP graph of its own, replayed BETWEEN the two halves of the
            # matching: the cost blocks and their copy to the host are enqueued first, the teacher's ~2 ms of GPU work run while the host
            # waits for that copy and solves the seven assignments (scipy) -- the round trip costs the step nothing
            for _ in range(2):
                model.teacher_part(images)
            torch.cuda.synchronize()
            teacher_graph = torch.cuda.CUDAGraph()
            with quiet_gc(), torch.cuda.graph(teacher_graph, stream=side):
                t_static = model.teacher_part(images)
            params = [p for p in model.parameters() if p.requires_grad]
            opt = make_optimizer(params) if optimizer else None
            last = {}

            def step():
                for p in params:
                    p.grad = None
                outs = ga(images)
                with torch.no_grad():
                    pending = model.match_begin(*outs[:4], targets)           # device cost blocks -> one host copy, enqueued
                    teacher_graph.replay()
                    assign = model.match_end(pending)                         # wait for the copy, scipy (matcher.py) -- under the teacher
                last["assign"] = assign
                loss = gb(*outs, t_static, *model.pack_indices(assign, targets))
                loss.backward()
                if opt is not None:
                    optimizer_step(opt, params)
                return loss

            for _ in range(warmup):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                loss = step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
    finally:
        torch.cuda.graph.default_capture_stream = saved_capture_stream
    torch.cuda.current_stream().wait_stream(side)
    return {"what": "the same step with its two device-only parts (model up to the matcher; criterion) captured forward + backward by "
                    "torch.cuda.make_graphed_callables and the Hungarian assignment live on the host between them every step, under the frozen "
                    "teacher's forward (a HIP graph of its own, replayed while the host waits for the cost blocks and solves the assignments)"
                    + ("; then gradient clipping (0.1) + fused AdamW step (reference engine.py:105-113): a TRAINING step, the same thing "
                       "full_step_ddp measures at N > 1" if optimizer else "; no optimizer step"),
            "optimizer": "AdamW(fused) + clip_grad_norm_(0.1)" if optimizer else None,
            "ms": round(ms, 2), "img_per_s": round(n_img / (ms * 1e-3), 2), "loss": float(loss.detach()),
            "grad_norm": float(torch.sqrt(sum((p.grad.float() ** 2).sum() for p in params if p.grad is not None))),
            **({"grads": {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None},
                "indices": last["assign"], "topk": model.last_topk.clone()} if return_grads else {})}


def run_ddp(n_img, dev, dist, steps=5, warmup=2, optimizer=True, make_model=None, backend_device=None):
    """The composed step as a data-parallel TRAINING step (round-3 verdict item 5; reference main.py:204-206 wraps the whole model in
    DistributedDataParallel, engine.py:100-114 runs backward + optimizer step): the module in ``DistributedDataParallel`` (nccl = RCCL
    on GPUs; ``gradient_as_bucket_view=True``, DDP's own 25 MB buckets overlapped with the backward), AdamW on every trained parameter,
    every rank on its own images.  Called by ALL ranks.  Returns, on every rank, the dict bench.py attaches as ``full_step_ddp``:
    ``ms`` (max over ranks, barrier + synchronise on both sides of the timed steps), ``ms_no_collective`` (the same steps under
    ``no_sync()``: no all-reduce) and the job's images per second.  ``make_model``: a stand-in module factory (tests/test_dist_gloo.py
    runs the protocol on CPU with gloo; the module needs ``batch()`` -> forward arguments and optionally ``prepare(*batch()[1:])``)."""
    import contextlib
    from torch.nn.parallel import DistributedDataParallel as DDP
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    model = make_model() if make_model is not None else Step(n_img=n_img, seed=0, dev=dev)      # (same weights on every rank; DDP broadcasts anyway)
    if hasattr(model, "timing"):
        model.timing = False
    rank = dist.get_rank() if world > 1 or (dist is not None and dist.is_initialized()) else 0
    batch = model.batch(seed=rank) if make_model is None else model.batch()
    if make_model is None:
        model.prepare(batch[1], batch[2])
    is_cuda = torch.device(dev).type == "cuda"
    if dist is not None and dist.is_initialized():
        ddp = DDP(model, device_ids=[torch.device(dev).index] if is_cuda else None, gradient_as_bucket_view=True, broadcast_buffers=False)
    else:
        ddp = model
    params = [p for p in model.parameters() if p.requires_grad]
    opt = (torch.optim.AdamW(params, lr=1e-4, weight_decay=1e-4, fused=True) if is_cuda else
           torch.optim.AdamW(params, lr=1e-4, weight_decay=1e-4)) if optimizer else None      # main.py:213-214

    def step(sync=True):
        if opt is not None:
            opt.zero_grad(set_to_none=True)
        else:
            for p in params:
                p.grad = None
        ctx = ddp.no_sync() if (not sync and ddp is not model) else contextlib.nullcontext()
        with ctx:
            loss = ddp(*batch)
            loss.backward()
        if opt is not None:
            torch.nn.utils.clip_grad_norm_(params, CLIP_MAX_NORM)      # engine.py:110-112
            opt.step()
        return loss

    def fence():
        if dist is not None and dist.is_initialized():
            dist.barrier()
        if is_cuda:
            torch.cuda.synchronize()

    def timed(sync):
        for _ in range(warmup):
            step(sync)
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step(sync)
        fence()
        el = time.perf_counter() - t0
        if dist is not None and dist.is_initialized() and world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=backend_device or (dev if is_cuda else "cpu"))
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el / steps * 1e3, float(loss.detach())

    ms, loss = timed(True)
    ms_nc, _ = timed(False)
    unused = [n for n, p in model.named_parameters() if p.requires_grad and p.grad is None]
    return {"what": "the composed step as a data-parallel training step: the module in DistributedDataParallel (RCCL, gradient_as_bucket_view, "
                    "25 MB buckets overlapped with the backward), gradient clipping (0.1) + AdamW step included, every rank on its own images; "
                    "ms = max over ranks",
            "world": world, "ms": round(ms, 2), "ms_no_collective": round(ms_nc, 2), "img_per_s": round(world * n_img / (ms * 1e-3), 2),
            "img_per_s_per_rank": round(n_img / (ms * 1e-3), 2), "optimizer": "AdamW" if optimizer else None, "loss": loss,
            "trained_parameters": sum(p.numel() for p in params), "parameters_without_gradient": unused}


if __name__ == "__main__":
    import argparse
    import json
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--images", type=int, default=2)
    ap.add_argument("--no-graph", action="store_true", help="eager steps only (the form profiled for profiles/*_step_kernels.md)")
    ap.add_argument("--stop-at", default=None, help="profiling aid: cut the step off after this section (implies --no-graph)")
    a_ = ap.parse_args()
    print(json.dumps(run(a_.images, torch.device("cuda", 0), a_.steps, a_.warmup, graph=not (a_.no_graph or a_.stop_at), stop_at=a_.stop_at), indent=1))
