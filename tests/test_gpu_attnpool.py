"""GPU (-m gpu): AttentionPool2d on the restructured single-query attention (msda_attnpool_core_*)
against the fixtures generated from the reference's class (clip/model.py:58-91) and, at CLIP-RN50's own size, against the oracle;
clip_box_targets (richsem.py:745-761) against the same steps written with the oracles."""
import glob
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import attnpool_oracle as AO      # noqa: E402
from oracle import roi_oracle as RO           # noqa: E402

pytestmark = pytest.mark.gpu
CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "attnpool_*.npz")))


def build(params, heads, spacial_dim, dtype):
    from richsem_amd.modules import AttentionPool2d
    C = params["positional_embedding"].shape[1]
    m = AttentionPool2d(spacial_dim, C, heads, params["c_proj.weight"].shape[0]).to(dtype)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})        # the reference's own state-dict keys
    return m.cuda().eval()


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_golden_vectors(path):
    x, params, heads, want = AO.load_case(path)
    dt = torch.float64 if x.dtype == np.float64 else torch.float32
    m = build(params, heads, x.shape[-1], dt)
    got = m(torch.from_numpy(x).cuda()).cpu().numpy()
    assert got.shape == want.shape
    tol = 1e-11 if x.dtype == np.float64 else 3e-5
    assert np.abs(got - want).max() <= tol * np.abs(want).max()
    assert "librichsem_msda.so" in open("/proc/self/maps").read()


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-11), (np.float32, 5e-5)])
def test_clip_rn50_size_against_oracle(dtype, tol):
    """spacial_dim 7, embed_dim 2048, 32 heads, output_dim 1024 (clip/model.py:126-127 for RN50), 24 ROIs (2 images x 12 boxes)"""
    rng = np.random.default_rng(0)
    C, H, sd, od, K = 2048, 32, 7, 1024, 24
    params = {"positional_embedding": rng.normal(0, C ** -0.5, (sd * sd + 1, C))}
    for n, o in (("q", C), ("k", C), ("v", C), ("c", od)):
        params[n + "_proj.weight"] = rng.normal(0, C ** -0.5, (o, C))
        params[n + "_proj.bias"] = rng.normal(0, 0.5, (o,))
    params = {k: v.astype(dtype) for k, v in params.items()}
    x = rng.normal(0, 1, (K, C, sd, sd)).astype(dtype)
    want = AO.attnpool(x, params, H)
    m = build(params, H, sd, torch.float64 if dtype == np.float64 else torch.float32)
    got = m(torch.from_numpy(x).cuda()).cpu().numpy()
    assert np.abs(got - want).max() <= tol * np.abs(want).max()


def test_edges():
    from richsem_amd.modules import AttentionPool2d
    m = AttentionPool2d(3, 32, 4, 16).cuda()
    assert m(torch.zeros(0, 32, 3, 3, device="cuda")).shape == (0, 16)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        AttentionPool2d(3, 32, 4, 16)(torch.zeros(1, 32, 3, 3))
    with pytest.raises(AssertionError):
        m(torch.zeros(1, 32, 4, 4, device="cuda"))
    assert sorted(m.state_dict().keys()) == sorted(["positional_embedding"] + [f"{n}_proj.{p}" for n in "qkvc" for p in ("weight", "bias")])


def test_clip_box_targets_against_the_oracles():
    """richsem.py:745-761 step by step with the numpy oracles (ROIAlign, attention pool), float64"""
    from richsem_amd.modules import AttentionPool2d, clip_box_targets
    rng = np.random.default_rng(1)
    C, H, sd, od, classes = 64, 4, 7, 32, 50
    feats = rng.normal(0, 1, (2, C, 25, 42))
    m = AttentionPool2d(sd, C, H, od).double().cuda().eval()
    params = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    sizes = [3, 0 + 2]
    targets, rois = [], []
    for b, n in enumerate(sizes):
        boxes = np.concatenate([rng.uniform(0.2, 0.8, (n, 2)), rng.uniform(0.05, 0.4, (n, 2))], -1)
        size = np.array([800.0, 1333.0])
        targets.append({"boxes": torch.from_numpy(boxes).cuda(), "size": torch.from_numpy(size).cuda(),
                        "labels": torch.zeros(n, dtype=torch.int64, device="cuda")})
        sc = size[[1, 0, 1, 0]]
        xyxy = np.stack([boxes[:, 0] - 0.5 * boxes[:, 2], boxes[:, 1] - 0.5 * boxes[:, 3], boxes[:, 0] + 0.5 * boxes[:, 2],
                         boxes[:, 1] + 0.5 * boxes[:, 3]], -1) * sc
        rois.append(np.concatenate([np.full((n, 1), float(b)), xyxy], -1))
    rois = np.concatenate(rois)
    text = rng.normal(0, 1, (classes, od))
    logit_scale = np.log(1 / 0.07)
    prompts, logits = clip_box_targets(torch.from_numpy(feats).cuda(), targets, m, torch.from_numpy(text).cuda(), logit_scale)
    pooled = RO.roi_align(feats, rois, (sd, sd), 1.0 / 32, 0, True)
    p = AO.attnpool(pooled, params, H)
    p = p / np.linalg.norm(p, axis=-1, keepdims=True)
    te = text / np.linalg.norm(text, axis=-1, keepdims=True)
    lg = p @ te.T * np.exp(logit_scale)
    assert [len(q) for q in prompts] == sizes and [len(q) for q in logits] == sizes
    assert np.abs(torch.cat(prompts).cpu().numpy() - p).max() < 1e-11
    assert np.abs(torch.cat(logits).cpu().numpy() - lg).max() < 1e-9
