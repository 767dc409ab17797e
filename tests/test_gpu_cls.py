"""GPU (-m gpu): the two-stage selection score as one MFMA kernel (csrc/cls_mfma.hip) against the numpy
oracle of the reference's op sequence (oracle/cls_oracle.py; parity unpinned, see its header).  Tolerances: the oracle is evaluated
in fp64 on the same fp32 inputs; `parts = 2` with fp32 memory must sit at fp32 level (the reference's own fp32 chain differs from the
fp64 value by about as much), bf16 modes at bf16 level."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import cls_oracle as CO          # noqa: E402

pytestmark = pytest.mark.gpu


def make(seed, tokens, classes=1204, proj=1024):
    rng = np.random.default_rng(seed)
    mem = rng.normal(0, 1, (tokens, 256)).astype(np.float32)
    wp = rng.normal(0, proj ** -0.5, (proj, 256)).astype(np.float32)          # richsem.py:77: normal_(std = l_dim ** -0.5)
    text = rng.normal(0, 1, (classes, proj)).astype(np.float32)
    return mem, wp, text, np.float32(np.log(1 / 0.07))                         # CLIP's initial logit scale


def scorer(wp, text, ls, parts):
    from richsem_amd.two_stage import ClassScorer
    return ClassScorer(parts).prepare(torch.from_numpy(wp).cuda(), torch.from_numpy(text).cuda(), torch.tensor(float(ls)))


@pytest.mark.parametrize("tokens,classes", [(1000, 1204), (193, 1203), (48, 16), (7, 5), (4097, 80)])
def test_fp32_memory_split_product_is_at_fp32_level(tokens, classes):
    mem, wp, text, ls = make(tokens, tokens, classes)
    want = CO.max_logits(mem.astype(np.float64), wp.astype(np.float64), text.astype(np.float64), np.float64(ls))
    got = scorer(wp, text, ls, 2).max_logits(torch.from_numpy(mem).cuda()).cpu().numpy()
    assert got.shape == want.shape and got.dtype == np.float32
    ref32 = CO.max_logits(mem, wp, text, ls)            # the reference's own precision: fp32 ops
    err, err32 = np.abs(got - want).max(), np.abs(ref32 - want).max()
    spread = np.abs(want).max()
    assert err <= 2e-5 * spread, (err, err32, spread)
    assert "librichsem_msda.so" in open("/proc/self/maps").read()


def test_bf16_modes_are_at_bf16_level():
    mem, wp, text, ls = make(1, 2000)
    want = CO.max_logits(mem.astype(np.float64), wp.astype(np.float64), text.astype(np.float64), np.float64(ls))
    spread = np.abs(want).max()
    x16 = torch.from_numpy(mem).cuda().to(torch.bfloat16)
    want16 = CO.max_logits(x16.float().cpu().numpy().astype(np.float64), wp.astype(np.float64), text.astype(np.float64), np.float64(ls))
    got = scorer(wp, text, ls, 2).max_logits(x16).cpu().numpy()         # bf16 memory, exact weights: fp32-level on the rounded input
    assert np.abs(got - want16).max() <= 2e-5 * spread
    got1 = scorer(wp, text, ls, 1).max_logits(x16).cpu().numpy()        # plain bf16 product
    assert np.abs(got1 - want16).max() <= 2e-2 * spread
    got1f = scorer(wp, text, ls, 1).max_logits(torch.from_numpy(mem).cuda()).cpu().numpy()
    assert np.abs(got1f - want).max() <= 2e-2 * spread


def test_topk_proposals_at_the_training_shape():
    """bs 2, 22323 pixels, 1204 classes, 900 queries (deformable_transformer.py:370-372)"""
    mem, wp, text, ls = make(2, 2 * 22323)
    sc = scorer(wp, text, ls, 2)
    memory = torch.from_numpy(mem).cuda().view(2, 22323, 256)
    idx = sc.topk_proposals(memory, 900)
    scores = sc.max_logits(memory)
    assert idx.shape == (2, 900) and idx.dtype == torch.int64
    assert torch.equal(idx, torch.topk(scores, 900, dim=1)[1])          # exact on the kernel's own scores
    want = CO.max_logits(mem.astype(np.float64), wp.astype(np.float64), text.astype(np.float64), np.float64(ls)).reshape(2, -1)
    ref_idx = CO.topk(want, 900)
    for b in range(2):
        got, ref = set(idx[b].tolist()), set(ref_idx[b].tolist())
        kth = want[b][ref_idx[b][-1]]
        # the two selections may differ only among rows whose fp64 score is within the fp32-level error of the 900th
        for i in got ^ ref:
            assert abs(want[b][i] - kth) <= 4e-5 * np.abs(want[b]).max()
        assert len(got ^ ref) <= 4


def test_errors():
    from richsem_amd.two_stage import ClassScorer
    mem, wp, text, ls = make(3, 10, 20, 64)
    with pytest.raises(RuntimeError, match="prepare"):
        ClassScorer().max_logits(torch.zeros(1, 256, device="cuda"))
    sc = scorer(wp, text, ls, 2)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        sc.max_logits(torch.zeros(1, 256))
    with pytest.raises(RuntimeError, match="float32 / bfloat16"):
        sc.max_logits(torch.zeros(1, 256, device="cuda", dtype=torch.float64))
    assert sc.max_logits(torch.zeros(0, 256, device="cuda")).shape == (0,)


def test_scorer_equals_the_reference_classifier_fixture():
    """ClassScorer (one MFMA kernel + the top-k kernel) against tests/golden/cls_clipalign.npz, produced by the reference's own
    CLIPAlign.forward followed by .max(-1)[0] and torch.topk (deformable_transformer.py:368-372): scores to the fp32-level tolerance of
    the split-bf16 product, selections equal except among scores within that tolerance of the k-th"""
    from tests.test_oracle_cls import cases
    for tag, _, c in cases():
        mem = torch.from_numpy(c["memory"]).float().cuda()
        sc = scorer(c["proj_weight"].astype(np.float32), c["text_embed"].astype(np.float32), float(c["logit_scale"]), 2)
        got = sc.max_logits(mem).cpu().numpy()
        want = c["scores"].astype(np.float64)
        spread = np.abs(want).max()
        assert np.abs(got - want).max() <= 2e-5 * spread, tag
        k = c["topk"].shape[1]
        idx = sc.topk_proposals(mem, k).cpu().numpy()
        for b in range(idx.shape[0]):
            a, r = set(idx[b].tolist()), set(c["topk"][b].tolist())
            kth = want[b][c["topk"][b][-1]]
            for i in a ^ r:
                assert abs(want[b][i] - kth) <= 4e-5 * spread
            assert len(a ^ r) <= 4
