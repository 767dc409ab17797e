"""CPU: the oracle (oracle/msda_oracle.c) against the golden vectors made from the reference's own
pure-PyTorch path (tests/golden/make_golden.py).  This is what pins the oracle."""
import glob
import os

import numpy as np
import pytest

from oracle import msda_oracle as O

from conftest import GOLDEN, OP_CASES

CASES = OP_CASES


def rel_err(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))


def tol_for(dtype):
    # fp64: rounding-order differences only; fp32: same, at single precision
    return 1e-12 if dtype == np.float64 else 2e-5


def test_cases_present():
    assert len(CASES) >= 11, CASES


@pytest.mark.parametrize("case", CASES)
def test_forward_matches_reference(case):
    z = np.load(os.path.join(GOLDEN, case + ".npz"))
    out = O.forward(z["value"], z["shapes"], z["lsi"], z["loc"], z["aw"])
    assert out.dtype == z["out"].dtype and out.shape == z["out"].shape
    assert rel_err(out, z["out"]) < tol_for(out.dtype)


@pytest.mark.parametrize("case", CASES)
def test_backward_matches_reference_autograd(case):
    z = np.load(os.path.join(GOLDEN, case + ".npz"))
    gv, gl, ga = O.backward(z["value"], z["shapes"], z["lsi"], z["loc"], z["aw"], z["grad_out"])
    tol = tol_for(gv.dtype)
    assert rel_err(gv, z["grad_value"]) < tol
    assert rel_err(ga, z["grad_aw"]) < tol
    if case == "border_exact_fwd":
        # Samples EXACTLY on the acceptance border (h_im == -1, h_im == H, same for w): the reference CUDA
        # kernel drops them (ms_deform_im2col_cuda.cuh:285 uses strict inequalities) so their location gradient
        # is 0, while grid_sample keeps them with zero weight and a one-sided derivative.  Measure-zero set;
        # the oracle follows the kernel.  Everywhere else the location gradients must agree.
        H = z["shapes"][:, 0].astype(np.float64)[None, None, None, :, None]
        W = z["shapes"][:, 1].astype(np.float64)[None, None, None, :, None]
        h_im = z["loc"][..., 1] * H - 0.5
        w_im = z["loc"][..., 0] * W - 0.5
        on_border = (h_im == -1) | (h_im == H) | (w_im == -1) | (w_im == W)
        assert on_border.any() and not on_border.all()
        assert np.all(gl[on_border] == 0)
        keep = ~on_border
        assert np.abs(gl[keep] - z["grad_loc"][keep]).max() / np.abs(z["grad_loc"]).max() < tol
    else:
        assert rel_err(gl, z["grad_loc"]) < tol


@pytest.mark.parametrize("case", CASES)
def test_grid_sample_restatement_matches_reference(case):
    """oracle/msda_torch_oracle.py (the reference's own CPU path restated, timed by bench.py's cpu_baseline) against the fixtures the
    reference's function produced: same F.grid_sample formulation, so agreement is at rounding level in both dtypes"""
    import torch
    from oracle import msda_torch_oracle as T
    z = np.load(os.path.join(GOLDEN, case + ".npz"))
    t = {k: torch.from_numpy(z[k]) for k in ("value", "shapes", "loc", "aw", "grad_out")}
    out, gv, gl, ga = T.forward_backward(t["value"], t["shapes"], t["loc"], t["aw"], t["grad_out"])
    tol = 1e-12 if z["value"].dtype == np.float64 else 1e-5
    assert rel_err(out.numpy().reshape(z["out"].shape), z["out"]) < tol
    assert rel_err(gv.numpy(), z["grad_value"]) < tol
    assert rel_err(gl.numpy(), z["grad_loc"]) < tol
    assert rel_err(ga.numpy(), z["grad_aw"]) < tol


def test_reference_test_tolerances():
    """The reference's own acceptance criteria (ops/test.py:40 default allclose in fp64; :56 rtol 1e-2, atol 1e-3
    in fp32) hold for the oracle on the reference's own inputs."""
    z = np.load(os.path.join(GOLDEN, "ref_test_fwd_double.npz"))
    assert np.allclose(O.forward(z["value"], z["shapes"], z["lsi"], z["loc"], z["aw"]), z["out"])
    z = np.load(os.path.join(GOLDEN, "ref_test_fwd_float.npz"))
    assert np.allclose(O.forward(z["value"], z["shapes"], z["lsi"], z["loc"], z["aw"]), z["out"], rtol=1e-2, atol=1e-3)


def test_threads_do_not_change_results():
    z = np.load(os.path.join(GOLDEN, "pyramid_encoder_f32.npz"))
    args = (z["value"], z["shapes"], z["lsi"], z["loc"], z["aw"])
    O.set_threads(1)
    o1 = O.forward(*args)
    g1 = O.backward(*args, z["grad_out"])
    O.set_threads(4)
    o4 = O.forward(*args)
    g4 = O.backward(*args, z["grad_out"])
    O.set_threads(1)
    assert np.array_equal(o1, o4)
    for a, b in zip(g1, g4):
        assert np.array_equal(a, b)


def test_oracle_adjoint_identity():
    """<out(value), g> == <value, grad_value(g)>: forward is linear in value and backward is its adjoint."""
    z = np.load(os.path.join(GOLDEN, "decoder_n2m8_f64.npz"))
    out = O.forward(z["value"], z["shapes"], z["lsi"], z["loc"], z["aw"])
    gv, _, _ = O.backward(z["value"], z["shapes"], z["lsi"], z["loc"], z["aw"], z["grad_out"])
    lhs = float((out * z["grad_out"]).sum())
    rhs = float((z["value"] * gv).sum())
    assert abs(lhs - rhs) < 1e-10 * max(1.0, abs(lhs))


def test_dn_oracle_matches_the_reference_prepare_for_cdn():
    """oracle/dn_oracle.py against outputs of the reference's own prepare_for_cdn (tests/golden/make_golden_layers.py): the
    attention mask, pad_size, the group count and the slots of the padded query tensors it fills, bit for bit -- ragged batches,
    an image without boxes, use_cdn on / off, add_gt, group counts that round to one"""
    from oracle import dn_oracle
    z = np.load(os.path.join(GOLDEN, "dn_prepare_for_cdn.npz"))
    n = len([k for k in z.files if k.endswith(".counts")])
    assert n >= 8
    for ci in range(n):
        pre = f"c{ci}."
        counts = [int(c) for c in z[pre + "counts"]]
        dn_number, use_cdn, add_gt, nq = (int(v) for v in z[pre + "args"])
        got = dn_oracle.prepare_for_cdn_indices(counts, dn_number, nq, bool(use_cdn), bool(add_gt))
        assert [got["pad_size"], got["num_dn_group"]] == z[pre + "meta"].tolist(), ci
        assert got["attn_mask"].shape == z[pre + "attn_mask"].shape and np.array_equal(got["attn_mask"], z[pre + "attn_mask"]), ci
        assert list(z[pre + "query_shape"]) == [len(counts), got["pad_size"], 4], ci
        assert np.array_equal(got["filled"], z[pre + "filled"]), ci
