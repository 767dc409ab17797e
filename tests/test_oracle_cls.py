"""CPU: the two-stage class-score oracle (oracle/cls_oracle.py) against the fixture generated from the reference's own CLIPAlign.forward
(tests/golden/make_golden_cls.py -> tests/golden/cls_clipalign.npz)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import cls_oracle as CO   # noqa: E402

FIX = os.path.join(ROOT, "tests", "golden", "cls_clipalign.npz")


def cases():
    z = np.load(FIX)
    for tag, tol in (("f64", 1e-12), ("f32", 2e-5)):
        yield tag, tol, {k.split(".", 1)[1]: z[k] for k in z.files if k.startswith(tag + ".")}


@pytest.mark.parametrize("tag,tol,c", list(cases()), ids=["f64", "f32"])
def test_oracle_equals_the_reference_classifier(tag, tol, c):
    logits = CO.class_logits(c["memory"], c["proj_weight"], c["text_embed"], c["logit_scale"])
    assert logits.dtype == c["scores"].dtype
    spread = np.abs(c["scores"]).max()
    assert np.abs(logits[:, :8] - c["logits_head"]).max() <= tol * spread
    scores = CO.max_logits(c["memory"], c["proj_weight"], c["text_embed"], c["logit_scale"])
    assert np.abs(scores - c["scores"]).max() <= tol * spread
    k = c["topk"].shape[1]
    got = CO.topk(c["scores"], k)                       # on the reference's own scores: the selection itself must be identical
    assert np.array_equal(got, c["topk"])
