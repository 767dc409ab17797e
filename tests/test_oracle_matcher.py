"""CPU: the matcher oracle (oracle/matcher_oracle.py) against the reference's formula written with torch CPU ops
(models/richsem/matcher.py:52-74; util/box_ops.py:9-59 with torchvision's box_area = (x1 - x0) * (y1 - y0)), and the host-side
criterion plumbing (global_num_boxes, AsyncLossLog) on one rank and on two gloo ranks."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import matcher_oracle as MO          # noqa: E402


def make_case(seed, bs=2, nq=37, C=50, sizes=(5, 3), dtype=np.float64):
    rng = np.random.default_rng(seed)
    logits = rng.normal(0, 2, (bs, nq, C)).astype(dtype)
    boxes = np.concatenate([rng.uniform(0.2, 0.8, (bs, nq, 2)), rng.uniform(0.05, 0.4, (bs, nq, 2))], -1).astype(dtype)
    labels = [rng.integers(0, C, s).astype(np.int64) for s in sizes]
    tboxes = [np.concatenate([rng.uniform(0.2, 0.8, (s, 2)), rng.uniform(0.05, 0.4, (s, 2))], -1).astype(dtype) for s in sizes]
    return logits, boxes, labels, tboxes


def torch_formula(logits, boxes, tgt_ids, tgt_boxes, wc, wb, wg, alpha):
    """matcher.py:52-74 with torch ops, box_ops restated line by line"""
    out_prob = torch.from_numpy(logits).flatten(0, 1).sigmoid()
    out_bbox = torch.from_numpy(boxes).flatten(0, 1)
    tgt_ids, tgt_bbox = torch.from_numpy(tgt_ids), torch.from_numpy(tgt_boxes)
    neg = (1 - alpha) * (out_prob ** 2.0) * (-(1 - out_prob + 1e-8).log())
    pos = alpha * ((1 - out_prob) ** 2.0) * (-(out_prob + 1e-8).log())
    cost_class = pos[:, tgt_ids] - neg[:, tgt_ids]
    cost_bbox = torch.cdist(out_bbox, tgt_bbox, p=1)

    def xyxy(x):
        xc, yc, w, h = x.unbind(-1)
        return torch.stack([xc - 0.5 * w, yc - 0.5 * h, xc + 0.5 * w, yc + 0.5 * h], dim=-1)

    b1, b2 = xyxy(out_bbox), xyxy(tgt_bbox)
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    lt, rb = torch.max(b1[:, None, :2], b2[:, :2]), torch.min(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    union = a1[:, None] + a2 - inter
    iou = inter / (union + 1e-6)
    lt, rb = torch.min(b1[:, None, :2], b2[:, :2]), torch.max(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    area = wh[:, :, 0] * wh[:, :, 1]
    giou = iou - (area - union) / (area + 1e-6)
    C = wb * cost_bbox + wc * cost_class + wg * (-giou)
    return C.view(logits.shape[0], logits.shape[1], -1).numpy()


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-12), (np.float32, 2e-5)])
def test_oracle_equals_torch_formula(dtype, tol):
    for seed in range(4):
        logits, boxes, labels, tboxes = make_case(seed, dtype=dtype)
        ids, tb = np.concatenate(labels), np.concatenate(tboxes)
        want = torch_formula(logits, boxes, ids, tb, 2.0, 5.0, 2.0, 0.25)
        got = MO.cost_matrix(logits, boxes, ids, tb, cost_class=2.0, cost_bbox=5.0, cost_giou=2.0, focal_alpha=0.25)
        assert got.dtype == dtype and got.shape == want.shape
        assert np.abs(got - want).max() <= tol * np.abs(want).max()


def test_oracle_match_shapes_and_optimality():
    logits, boxes, labels, tboxes = make_case(7, sizes=(6, 0))
    res = MO.match(logits, boxes, labels, tboxes, cost_class=2.0, cost_bbox=5.0, cost_giou=2.0)
    assert [len(i) for i, _ in res] == [6, 0]
    i, j = res[0]
    assert sorted(j.tolist()) == list(range(6)) and len(set(i.tolist())) == 6


def test_num_boxes_and_loss_log_single_rank():
    from richsem_amd.matcher import AsyncLossLog, global_num_boxes
    idx = [(torch.arange(3), torch.arange(3)), (torch.arange(0), torch.arange(0))]
    assert global_num_boxes(idx) == 3.0
    assert global_num_boxes([(torch.arange(0), torch.arange(0))]) == 1.0          # clamp(min = 1), richsem.py:1147
    log = AsyncLossLog()
    assert log.push({"loss_b": torch.tensor(2.0), "loss_a": torch.tensor(1.0)}) is None
    prev = log.push({"loss_b": torch.tensor(4.0), "loss_a": torch.tensor(3.0)})
    assert prev == {"loss_a": 1.0, "loss_b": 2.0}
    assert log.flush() == {"loss_a": 3.0, "loss_b": 4.0} and log.flush() is None


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from richsem_amd.matcher import AsyncLossLog, global_num_boxes
    idx = [(torch.arange(2 + 3 * rank), torch.arange(2 + 3 * rank))]
    assert global_num_boxes(idx, world_size=world) == (2 + 5) / 2            # sum over ranks / world size (richsem.py:1143-1147)
    log = AsyncLossLog(world_size=world)
    log.push({"loss_ce": torch.tensor(1.0 + rank), "loss_bbox": torch.tensor(10.0 * (rank + 1))})
    got = log.flush()
    assert got == {"loss_bbox": 15.0, "loss_ce": 1.5}, got                    # averaged as util/misc.py:160-162
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_num_boxes_and_loss_log_two_gloo_ranks():
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, _free_port()), nprocs=2, join=True)


def _fixture_cases():
    z = np.load(os.path.join(ROOT, "tests", "golden", "matcher_hungarian.npz"))
    for tag, tol in (("f64", 1e-12), ("f32", 2e-5)):
        sizes = z[f"{tag}.sizes"].tolist()
        offs = np.concatenate(([0], np.cumsum(sizes)))
        labels = [z[f"{tag}.labels"][offs[b]:offs[b + 1]] for b in range(len(sizes))]
        tboxes = [z[f"{tag}.tgt_boxes"][offs[b]:offs[b + 1]] for b in range(len(sizes))]
        blocks = [z[f"{tag}.block{b}"] for b in range(len(sizes))]
        idx = [(z[f"{tag}.idx_i{b}"], z[f"{tag}.idx_j{b}"]) for b in range(len(sizes))]
        yield tag, tol, z[f"{tag}.logits"], z[f"{tag}.boxes"], labels, tboxes, offs, blocks, idx


def test_oracle_equals_the_reference_matcher_fixture():
    """oracle/matcher_oracle.py against tests/golden/matcher_hungarian.npz: the cost blocks the REFERENCE's HungarianMatcher handed to
    scipy and the assignments it returned (tests/golden/make_golden_matcher.py: the reference's matcher.py and box_ops functions,
    executed with torchvision's box_area restated) -- ragged targets, an image without targets, fp64 and fp32"""
    for tag, tol, logits, boxes, labels, tboxes, offs, blocks, idx in _fixture_cases():
        full = MO.cost_matrix(logits, boxes, np.concatenate(labels), np.concatenate(tboxes), cost_class=2.0, cost_bbox=5.0, cost_giou=2.0,
                              focal_alpha=0.25)
        for b, want in enumerate(blocks):
            got = full[b][:, offs[b]:offs[b + 1]]
            assert got.shape == want.shape and got.dtype == want.dtype
            if want.size:
                assert np.abs(got - want).max() <= tol * max(np.abs(want).max(), 1.0), tag
        res = MO.match(logits, boxes, labels, tboxes, cost_class=2.0, cost_bbox=5.0, cost_giou=2.0)
        for (gi, gj), (wi, wj) in zip(res, idx):
            assert gi.tolist() == wi.tolist() and gj.tolist() == wj.tolist(), tag
