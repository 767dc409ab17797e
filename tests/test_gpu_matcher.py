"""GPU (-m gpu): the matcher's cost blocks (msda_matcher_cost_*) against the numpy oracle
(oracle/matcher_oracle.py: reference models/richsem/matcher.py:49-78; pinned by tests/golden/matcher_hungarian.npz, see its header), that fixture, and the mirror class's
assignments against the oracle's."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import matcher_oracle as MO          # noqa: E402
from tests.test_oracle_matcher import make_case  # noqa: E402

pytestmark = pytest.mark.gpu

W = dict(cost_class=2.0, cost_bbox=5.0, cost_giou=2.0)     # config/RichSem/richsem_4scale.py: set_cost_class / bbox / giou


def to_targets(labels, tboxes, dev, dtype):
    return [{"labels": torch.from_numpy(l).to(dev), "boxes": torch.from_numpy(b).to(dev, dtype)} for l, b in zip(labels, tboxes)]


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-12), (np.float32, 2e-5)])
@pytest.mark.parametrize("sizes", [(5, 3), (12, 12), (0, 4), (1, 0), (40, 7)])
def test_cost_blocks_equal_oracle(dtype, tol, sizes):
    from richsem_amd.matcher import CostPlan, cost_blocks
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    logits, boxes, labels, tboxes = make_case(11, nq=93, C=120, sizes=sizes, dtype=dtype)
    plan = CostPlan(to_targets(labels, tboxes, "cuda", tdt), torch.device("cuda"), tdt)
    got = cost_blocks(torch.from_numpy(logits).cuda(), torch.from_numpy(boxes).cuda(), plan, W["cost_class"], W["cost_bbox"],
                      W["cost_giou"], 0.25).cpu().numpy()
    full = MO.cost_matrix(logits, boxes, np.concatenate(labels), np.concatenate(tboxes), focal_alpha=0.25, **W)
    nq, t0 = logits.shape[1], 0
    for b, s in enumerate(sizes):
        block = got[nq * t0: nq * (t0 + s)].reshape(nq, s)
        want = full[b][:, t0:t0 + s]
        assert np.abs(block - want).max() <= tol * max(np.abs(want).max(), 1.0) if s else block.size == 0
        t0 += s


def test_forward_equals_oracle_assignment_at_the_training_shape():
    """bs 2, 900 queries, 1203 classes, 12 boxes per image"""
    from richsem_amd.matcher import HungarianMatcher
    logits, boxes, labels, tboxes = make_case(3, nq=900, C=1203, sizes=(12, 12), dtype=np.float32)
    m = HungarianMatcher(**W, focal_alpha=0.25)
    out = {"pred_logits": torch.from_numpy(logits).cuda(), "pred_boxes": torch.from_numpy(boxes).cuda()}
    got = m(out, to_targets(labels, tboxes, "cuda", torch.float32))
    want = MO.match(logits, boxes, labels, tboxes, **W)
    for (gi, gj), (wi, wj) in zip(got, want):
        assert gi.dtype == torch.int64 and gj.dtype == torch.int64
        assert gi.tolist() == wi.tolist() and gj.tolist() == wj.tolist()


def test_match_many_equals_one_by_one_with_one_host_copy():
    from richsem_amd.matcher import HungarianMatcher
    m = HungarianMatcher(**W)
    outs, cases = [], []
    _, _, labels, tboxes = make_case(0, nq=1, C=80, sizes=(7, 0, 9), bs=3, dtype=np.float32)
    for o, nq in enumerate([900, 900, 900, 900, 900, 900, 900, 200]):       # 6 decoder layers + the intermediate output + a DN-sized one
        logits, boxes, _, _ = make_case(100 + o, bs=3, nq=nq, C=80, sizes=(7, 0, 9), dtype=np.float32)
        cases.append((logits, boxes))
        outs.append({"pred_logits": torch.from_numpy(logits).cuda(), "pred_boxes": torch.from_numpy(boxes).cuda()})
    targets = to_targets(labels, tboxes, "cuda", torch.float32)
    many = m.match_many(outs, targets)
    assert len(many) == len(outs)
    for o, (logits, boxes) in enumerate(cases):
        want = MO.match(logits, boxes, labels, tboxes, **W)
        one = m(outs[o], targets)
        for (gi, gj), (si, sj), (wi, wj) in zip(many[o], one, want):
            assert gi.tolist() == wi.tolist() == si.tolist() and gj.tolist() == wj.tolist() == sj.tolist()


def test_errors_and_edges():
    from richsem_amd.matcher import CostPlan, HungarianMatcher, cost_blocks
    logits, boxes, labels, tboxes = make_case(5, dtype=np.float32)
    m = HungarianMatcher(**W)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        m({"pred_logits": torch.from_numpy(logits), "pred_boxes": torch.from_numpy(boxes)}, to_targets(labels, tboxes, "cpu", torch.float32))
    with pytest.raises(AssertionError):
        HungarianMatcher(0, 0, 0)
    # no targets at all: empty index pairs, nothing launched
    empty = [{"labels": torch.zeros(0, dtype=torch.int64, device="cuda"), "boxes": torch.zeros(0, 4, device="cuda")} for _ in range(2)]
    res = m({"pred_logits": torch.from_numpy(logits).cuda(), "pred_boxes": torch.from_numpy(boxes).cuda()}, empty)
    assert all(len(i) == 0 and len(j) == 0 for i, j in res)
    # a label outside [0, C): NaN column (the reference raises an index error)
    labels[0][1] = 10 ** 6
    plan = CostPlan(to_targets(labels, tboxes, "cuda", torch.float32), torch.device("cuda"), torch.float32)
    got = cost_blocks(torch.from_numpy(logits).cuda(), torch.from_numpy(boxes).cuda(), plan, 2.0, 5.0, 2.0, 0.25).cpu().numpy()
    block = got[: logits.shape[1] * len(labels[0])].reshape(logits.shape[1], -1)
    assert np.isnan(block[:, 1]).all() and not np.isnan(np.delete(block, 1, axis=1)).any()


def test_mirror_equals_the_reference_matcher_fixture():
    """richsem_amd.matcher.HungarianMatcher on the GPU against tests/golden/matcher_hungarian.npz: the cost blocks the reference's
    HungarianMatcher handed to scipy (to 1e-12 / 2e-5) and the assignments it returned (equal)"""
    from richsem_amd.matcher import CostPlan, HungarianMatcher, cost_blocks
    from tests.test_oracle_matcher import _fixture_cases
    for tag, tol, logits, boxes, labels, tboxes, offs, blocks, idx in _fixture_cases():
        tdt = torch.float64 if tag == "f64" else torch.float32
        targets = to_targets(labels, tboxes, "cuda", tdt)
        plan = CostPlan(targets, torch.device("cuda"), tdt)
        got = cost_blocks(torch.from_numpy(logits).cuda(), torch.from_numpy(boxes).cuda(), plan, W["cost_class"], W["cost_bbox"],
                          W["cost_giou"], 0.25).cpu().numpy()
        nq = logits.shape[1]
        for b, want in enumerate(blocks):
            block = got[nq * offs[b]: nq * offs[b + 1]].reshape(nq, -1)
            assert block.shape == want.shape
            if want.size:
                assert np.abs(block - want).max() <= tol * max(np.abs(want).max(), 1.0), tag
        res = HungarianMatcher(**W, focal_alpha=0.25)({"pred_logits": torch.from_numpy(logits).cuda(), "pred_boxes": torch.from_numpy(boxes).cuda()},
                                                     targets)
        for (gi, gj), (wi, wj) in zip(res, idx):
            assert gi.tolist() == wi.tolist() and gj.tolist() == wj.tolist(), tag
