"""GPU (-m gpu): the integer part of the denoising set-up (reference dn_components.py:27-61,
131-179) -- device tensors of richsem_amd/dn.py against the numpy restatement oracle/dn_oracle.py and against outputs of the
reference's own prepare_for_cdn (tests/golden/dn_prepare_for_cdn.npz), BIT-EXACT (int64 / bool)."""
import numpy as np
import pytest
import torch

from oracle import dn_oracle
from richsem_amd import _lib
from richsem_amd.dn import dn_group_count, prepare_dn_layout

pytestmark = pytest.mark.gpu

CASES = [
    ([12, 12], 100, 900, True),          # the bench's synthetic batch: 12 boxes per image
    ([3, 0, 7, 1], 100, 900, True),      # ragged, an image without boxes
    ([1], 100, 300, True),
    ([5, 9], 100, 900, False),           # use_cdn = False: negative halves dropped
    ([40, 13, 27], 100, 900, True),      # many boxes: few groups
    ([2, 2], 3, 50, True),               # small dn_number (not rescaled)
    ([60, 60], 100, 100, True),          # dn_number // (2 * max) == 0 -> 1 group
    ([0, 0], 100, 10, True),             # no boxes at all: one (empty) group
]


@pytest.mark.parametrize("known_num,dn_number,num_queries,use_cdn", CASES)
def test_layout_is_bit_exact(known_num, dn_number, num_queries, use_cdn):
    want = dn_oracle.prepare_for_cdn_indices(known_num, dn_number, num_queries, use_cdn)
    got = prepare_dn_layout(known_num, dn_number, num_queries, use_cdn)
    for k in ("pad_size", "num_dn_group", "single_pad", "group_pad"):
        assert got[k] == want[k], k
    for k in ("known_bid", "map_known_indice", "positive_idx", "negative_idx"):
        a = got[k].cpu().numpy()
        assert a.dtype == np.int64 and a.shape == want[k].shape and np.array_equal(a, want[k]), k
    m = got["attn_mask"]
    assert m.dtype == torch.bool and np.array_equal(m.cpu().numpy(), want["attn_mask"])


def test_layout_against_the_reference_fixture():
    """the device tensors against what the reference's own prepare_for_cdn returned (tests/golden/make_golden_layers.py)"""
    import os
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "dn_prepare_for_cdn.npz"))
    n = len([k for k in z.files if k.endswith(".counts")])
    for ci in range(n):
        pre = f"c{ci}."
        counts = [int(c) for c in z[pre + "counts"]]
        dn_number, use_cdn, add_gt, nq = (int(v) for v in z[pre + "args"])
        got = prepare_dn_layout(counts, dn_number, nq, bool(use_cdn), bool(add_gt))
        assert [got["pad_size"], got["num_dn_group"]] == z[pre + "meta"].tolist(), ci
        assert np.array_equal(got["attn_mask"].cpu().numpy(), z[pre + "attn_mask"]), ci
        full = got["single_pad"] * 2 * got["num_dn_group"]
        filled = torch.zeros((len(counts), full), dtype=torch.bool, device="cuda")
        if got["known_bid"].numel():
            filled[got["known_bid"], got["map_known_indice"]] = True       # dn_components.py:140-142
        if not use_cdn:
            filled = filled[:, got["positive_idx"]]                          # :145-149
        assert np.array_equal(filled.cpu().numpy(), z[pre + "filled"]), ci


def test_add_gt_and_group_count_follow_the_reference_arithmetic():
    for known, dn, add_gt in (([12, 12], 100, True), ([7], 100, False), ([0], 100, False), ([3], 0, False), ([200], 100, False)):
        want = dn_oracle.prepare_for_cdn_indices(known, dn, 10, True, add_gt)["num_dn_group"]
        assert dn_group_count(dn, known, add_gt) == want
    got = prepare_dn_layout([4, 2], 100, 20, True, add_gt=True)
    want = dn_oracle.prepare_for_cdn_indices([4, 2], 100, 20, True, add_gt=True)
    assert np.array_equal(got["attn_mask"].cpu().numpy(), want["attn_mask"])
    assert np.array_equal(got["map_known_indice"].cpu().numpy(), want["map_known_indice"])


def test_scatter_through_the_indices_reproduces_the_padded_layout():
    """dn_components.py:140-142: input_query[(known_bid, map_known_indice)] = values -- every (image, slot) pair is hit once"""
    got = prepare_dn_layout([3, 5], 100, 30, True)
    bid, slot = got["known_bid"], got["map_known_indice"]
    pairs = torch.stack([bid, slot], 1)
    assert pairs.unique(dim=0).shape[0] == pairs.shape[0]
    assert int(slot.max()) < got["pad_size"] and int(bid.max()) == 1


def test_bad_arguments():
    lib = _lib.load()
    m = torch.empty(16, dtype=torch.uint8, device="cuda")
    assert lib.msda_dn_attn_mask_u8(m.data_ptr(), 4, 5, 1, None) == -2     # pad_size > tgt_size
    assert lib.msda_dn_attn_mask_u8(None, 4, 2, 1, None) == -1
    with pytest.raises(RuntimeError):
        prepare_dn_layout([1], 100, 10, device="cpu")


# ---- top-k query selection (deformable_transformer.py:370-372) ---------------------------------------------------------------
from richsem_amd.dn import topk_indices   # noqa: E402


@pytest.mark.parametrize("rows,n,k", [(2, 22323, 900), (2, 34000, 900), (1, 1, 1), (3, 1000, 1000), (4, 5000, 1), (2, 36864, 1024),
                                      (5, 777, 300)])
def test_topk_equals_torch_topk_on_distinct_scores(rows, n, k):
    g = torch.Generator(device="cuda").manual_seed(rows * 1000 + k)
    # pairwise different scores per row (torch.topk leaves the order of equal scores open): a random permutation of a grid
    scores = torch.stack([torch.randperm(n, device="cuda", generator=g).float() for _ in range(rows)]) * (8.0 / n) - 4.0
    assert all(scores[r].unique().numel() == n for r in range(rows))
    idx, val = topk_indices(scores, k, return_values=True)
    tv, ti = torch.topk(scores, k, dim=1)
    assert idx.dtype == torch.int64 and torch.equal(idx, ti) and torch.equal(val, tv)


def test_topk_with_ties_infinities_and_negative_zero():
    scores = torch.zeros(2, 3000, device="cuda")
    scores[0, 100:160] = 5.0                      # 60 equal maxima, then zeros (+0.0 and -0.0 are different keys: -0.0 sorts below)
    scores[0, 7] = float("inf")
    scores[0, 9] = -float("inf")
    scores[1] = torch.arange(3000, device="cuda").float().remainder(7)   # heavy ties
    idx, val = topk_indices(scores, 64, return_values=True)
    tv, _ = torch.topk(scores, 64, dim=1)
    assert torch.equal(val, tv)                                           # same multiset of scores, descending
    assert idx[0, 0] == 7 and torch.equal(idx[0, 1:61], torch.arange(100, 160, device="cuda"))
    assert torch.equal(idx[0, 61:64], torch.tensor([0, 1, 2], device="cuda"))          # ties: lowest index first
    exp = torch.cat([torch.arange(6, 3000, 7)[:64]]).cuda()
    assert torch.equal(idx[1], exp[:64])
    assert all(idx[r].unique().numel() == 64 for r in range(2))


def test_topk_bad_arguments_and_fallback():
    lib = _lib.load()
    s = torch.zeros(1, 10, device="cuda")
    i = torch.empty(1, 20, dtype=torch.int64, device="cuda")
    assert lib.msda_topk_f32(s.data_ptr(), 1, 10, 11, i.data_ptr(), None, None) == -2
    big = torch.randn(1, 40000, device="cuda")
    assert torch.equal(topk_indices(big, 5), torch.topk(big, 5, dim=1)[1])          # too long for the kernel: torch.topk
