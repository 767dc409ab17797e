"""CPU: the attention-pool oracle against the fixtures generated from the reference's own class (clip/model.py:58-91 through
tests/golden/make_golden_attnpool.py)."""
import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import attnpool_oracle as AO      # noqa: E402

CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "attnpool_*.npz")))


def test_fixtures_present():
    assert len(CASES) == 4


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_oracle_reproduces_reference_output(path):
    x, params, heads, want = AO.load_case(path)
    got = AO.attnpool(x, params, heads)
    assert got.shape == want.shape and got.dtype == want.dtype
    tol = 1e-12 if x.dtype == np.float64 else 2e-5
    assert np.abs(got - want).max() <= tol * np.abs(want).max()
