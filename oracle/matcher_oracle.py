"""TEST INFRASTRUCTURE (oracle) -- numpy restatement of the reference's matching cost, never imported by the product.

Follows models/richsem/matcher.py:49-78 (HungarianMatcher.forward) and util/box_ops.py:9-59 (box_cxcywh_to_xyxy, box_iou,
generalized_box_iou; torchvision's box_area is (x1 - x0) * (y1 - y0)).  **PINNED** (round 3): tests/golden/matcher_hungarian.npz holds the
cost blocks the reference's own HungarianMatcher handed to scipy and the assignments it returned (tests/golden/make_golden_matcher.py
executes the reference's matcher.py and box_ops functions; the one third-party function they need, torchvision.ops.boxes.box_area --
torchvision is absent from the image --, is restated there from its published definition); tests/test_oracle_matcher.py holds this file
to them (1e-12 in fp64, 2e-5 in fp32, equal assignments) and, as before, to the same formula written with torch CPU ops.
"""
import numpy as np
from scipy.optimize import linear_sum_assignment


def cxcywh_to_xyxy(b):
    cx, cy, w, h = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], axis=-1)      # box_ops.py:9-13


def generalized_box_iou(b1, b2):
    dt = b1.dtype.type
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    lt = np.maximum(b1[:, None, :2], b2[None, :, :2])                                        # box_ops.py:29-30
    rb = np.minimum(b1[:, None, 2:], b2[None, :, 2:])
    wh = np.clip(rb - lt, 0, None)
    inter = wh[..., 0] * wh[..., 1]
    union = a1[:, None] + a2[None, :] - inter                                                # :35
    iou = inter / (union + dt(1e-6))
    lt = np.minimum(b1[:, None, :2], b2[None, :, :2])                                        # :56-57
    rb = np.maximum(b1[:, None, 2:], b2[None, :, 2:])
    wh = np.clip(rb - lt, 0, None)
    area = wh[..., 0] * wh[..., 1]
    return iou - (area - union) / (area + dt(1e-6))


def cost_matrix(logits, boxes, tgt_ids, tgt_boxes, cost_class=2.0, cost_bbox=5.0, cost_giou=2.0, focal_alpha=0.25):
    """logits (bs, nq, C), boxes (bs, nq, 4), tgt_ids (T), tgt_boxes (T, 4) -> the reference's FULL matrix (bs, nq, T)
    (matcher.py:52-74), in the dtype of `logits`."""
    dt = logits.dtype.type
    bs, nq = logits.shape[:2]
    x = logits.reshape(bs * nq, -1)
    p = dt(1) / (dt(1) + np.exp(-x))
    ob = boxes.reshape(bs * nq, 4)
    alpha = dt(focal_alpha)
    neg = (dt(1) - alpha) * (p * p) * (-np.log(dt(1) - p + dt(1e-8)))
    pos = alpha * ((dt(1) - p) * (dt(1) - p)) * (-np.log(p + dt(1e-8)))
    cc = pos[:, tgt_ids] - neg[:, tgt_ids]
    cb = np.abs(ob[:, None, :] - tgt_boxes[None, :, :]).sum(-1)                              # torch.cdist(p = 1)
    cg = -generalized_box_iou(cxcywh_to_xyxy(ob), cxcywh_to_xyxy(tgt_boxes))
    C = dt(cost_bbox) * cb + dt(cost_class) * cc + dt(cost_giou) * cg
    return C.reshape(bs, nq, -1)


def match(logits, boxes, labels_per_image, boxes_per_image, **weights):
    """The reference's forward: list over images of (query indices, target indices) (matcher.py:76-78)."""
    tgt_ids = np.concatenate([np.asarray(l, dtype=np.int64) for l in labels_per_image])
    tgt_boxes = np.concatenate([np.asarray(b, dtype=logits.dtype).reshape(-1, 4) for b in boxes_per_image])
    C = cost_matrix(logits, boxes, tgt_ids, tgt_boxes, **weights)
    out, t0 = [], 0
    for b, l in enumerate(labels_per_image):
        i, j = linear_sum_assignment(C[b][:, t0:t0 + len(l)])
        out.append((i.astype(np.int64), j.astype(np.int64)))
        t0 += len(l)
    return out
