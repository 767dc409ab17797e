"""GPU (-m gpu): ROIAlign forward (reference richsem.py:878) against the numpy restatement of
detectron2's published algorithm (oracle/roi_oracle.py: parity unpinned, see its header).  f64 to 1e-12, f32 to 1e-5."""
import numpy as np
import pytest
import torch

from oracle import roi_oracle
from richsem_amd import _lib
from richsem_amd.roi import ROIAlign, roi_align

pytestmark = pytest.mark.gpu


def _rois(rng, K, N, Hpx, Wpx):
    cx, cy = rng.uniform(0.2, 0.8, K) * Wpx, rng.uniform(0.2, 0.8, K) * Hpx
    w, h = rng.uniform(0.05, 0.4, K) * Wpx, rng.uniform(0.05, 0.4, K) * Hpx
    return np.stack([rng.integers(0, N, K).astype(np.float64), cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-12), (torch.float32, 1e-5)])
@pytest.mark.parametrize("shape,out,scale,ratio,aligned", [
    ((2, 6, 25, 42), 7, 1 / 32, 0, True),        # the reference's call: CLIP map of an 800 x 1344 image, 7 x 7 bins
    ((2, 3, 25, 42), 7, 1 / 32, 2, True),
    ((1, 4, 13, 9), (3, 5), 0.25, 0, False),     # legacy (not aligned) variant, rectangular output
    ((3, 2, 8, 8), 2, 1.0, 3, True),
])
def test_matches_the_restated_algorithm(shape, out, scale, ratio, aligned, dtype, tol):
    rng = np.random.default_rng(abs(hash((shape, ratio))) % 2 ** 31)
    N, C, H, W = shape
    inp = rng.standard_normal(shape)
    rois = _rois(rng, 9, N, H / scale, W / scale)
    rois[0, 1:] = [-50.0, -30.0, 40.0, 60.0]                      # sticks out of the map on the top left
    rois[1, 1:] = [W / scale - 20, H / scale - 20, W / scale + 90, H / scale + 70]   # ... and on the bottom right
    rois[2, 1:] = [10.0, 10.0, 10.5, 10.2]                        # smaller than a bin
    want = roi_oracle.roi_align(inp, rois, out, scale, ratio, aligned)
    got = roi_align(torch.from_numpy(inp).to(dtype).cuda(), torch.from_numpy(rois).to(dtype).cuda(), out, scale, ratio, aligned)
    assert got.shape == want.shape and got.dtype == dtype
    err = np.abs(got.double().cpu().numpy() - want).max() / (np.abs(want).max() + 1e-30)
    assert err < tol


def test_the_reference_call_shape_and_module_interface():
    x = torch.randn(2, 2048, 25, 42, device="cuda")
    rois = torch.tensor([[0, 100.0, 120.0, 500.0, 640.0], [1, 0.0, 0.0, 1344.0, 800.0]], device="cuda")
    out = ROIAlign(7, 1 / 32, 0, aligned=True).forward(x, rois)
    assert out.shape == (2, 2048, 7, 7) and torch.isfinite(out).all()
    # a box covering the whole map with one bin per ... averages to something close to the map's mean
    whole = ROIAlign(1, 1 / 32, 0, aligned=True)(x, rois[1:])
    assert float((whole[0, :, 0, 0] - x[1].mean((1, 2))).abs().max()) < 0.05
    assert ROIAlign(7, 1 / 32, 0)(x, rois[:0]).shape == (0, 2048, 7, 7)


def test_bad_arguments():
    lib = _lib.load()
    x = torch.zeros(1, 1, 4, 4, device="cuda")
    r = torch.zeros(1, 5, device="cuda")
    o = torch.zeros(1, 1, 2, 2, device="cuda")
    assert lib.msda_roi_align_forward_f32(x.data_ptr(), r.data_ptr(), 1, 1, 1, 4, 4, 0, 2, 1.0, 0, 1, o.data_ptr(), None) == -2
    assert lib.msda_roi_align_forward_f32(None, r.data_ptr(), 1, 1, 1, 4, 4, 2, 2, 1.0, 0, 1, o.data_ptr(), None) == -1
    with pytest.raises(RuntimeError):
        roi_align(x.cpu(), r.cpu(), 2)
