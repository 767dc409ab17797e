"""GPU (-m gpu): the CLIP ModifiedResNet teacher on the MFMA convolution kernel (richsem_amd/clip_resnet.py, SURVEY.md section 8a row
a11) against the fixtures generated from the reference's class (fp32) and against the oracle on another input.  Tolerance: bf16 storage
between ~16 layers with fp32 accumulation -- per-element error a few bf16 ulps of the map's scale, stated below; the same comparison
with the oracle run on bf16-rounded weights separates rounding of the weights from everything else."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import clip_resnet_oracle as RO                                  # noqa: E402
from clip_resnet_params import CASES                                        # noqa: E402
from test_oracle_clip_resnet import case_state_dict                         # noqa: E402

pytestmark = pytest.mark.gpu


def check(got, want, max_tol, mean_tol):
    scale = np.abs(want).max()
    err = np.abs(got - want)
    assert err.max() <= max_tol * scale, (err.max() / scale, err.mean() / scale)
    assert err.mean() <= mean_tol * scale, (err.max() / scale, err.mean() / scale)


@pytest.mark.parametrize("name", sorted(CASES))
def test_golden_vectors(name):
    from richsem_amd.clip_resnet import ModifiedResNetTeacher
    z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    sd = case_state_dict(name)
    m = ModifiedResNetTeacher(sd, heads=CASES[name][2])
    _, fmap = m(torch.from_numpy(z["x"]).cuda(), ret_sp=True)
    assert fmap.shape == z["fmap"].shape and fmap.dtype == torch.float32
    check(fmap.cpu().numpy(), z["fmap"], 4e-2, 4e-3)
    if "embed" in z.files:
        emb = m(torch.from_numpy(z["x"]).cuda())
        check(emb.cpu().numpy(), z["embed"], 4e-2, 8e-3)
    assert "librichsem_msda.so" in open("/proc/self/maps").read()


def test_against_oracle_with_rounded_weights_on_a_wide_image():
    """same network, 1 x 3 x 160 x 288 input; the oracle gets the bf16-rounded convolution weights the kernel uses, so what is left is
    the bf16 storage of the activations"""
    from richsem_amd.clip_resnet import ModifiedResNetTeacher
    name = "clip_resnet_w32_l2111"
    sd = case_state_dict(name)
    x = torch.from_numpy(np.random.default_rng(5).normal(0, 1, (1, 3, 160, 288)).astype(np.float32))
    m = ModifiedResNetTeacher(sd, heads=CASES[name][2])
    got = m(x.cuda(), ret_sp=True)[1].cpu().numpy()
    sd16 = {k: (v.to(torch.bfloat16).float() if v.dim() == 4 else v) for k, v in sd.items()}
    want = RO.feature_map(x.to(torch.bfloat16).float(), sd16).numpy()
    assert got.shape == want.shape == (1, 1024, 5, 9)
    check(got, want, 3e-2, 3e-3)
