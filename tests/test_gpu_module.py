"""GPU (-m gpu): the nn.Module mirror `MSDeformAttn` (row a7 of SURVEY.md section 8a) against golden vectors produced by the
reference's own module (reference models/richsem/ops/modules/ms_deform_attn.py:78-115) -- value projection with padding
mask, offsets / attention projections + softmax, 2-d and 4-d reference points, output projection -- forward, input
gradients and every parameter gradient.  fp64 end to end (the operator's f64 kernels), tolerance 1e-9 relative."""
import os

import numpy as np
import pytest
import torch

from richsem_amd.modules import MSDeformAttn

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.abs(a.detach().cpu().numpy() - b).max() / (np.abs(b).max() + 1e-300))


@pytest.mark.parametrize("case", ["module_encoder_ref2d", "module_decoder_ref4d"])
def test_module_matches_reference_module(case):
    z = np.load(os.path.join(GOLDEN, case + ".npz"))
    params = {k[len("param."):]: z[k] for k in z.files if k.startswith("param.") and not k.endswith(".grad")}
    C = params["value_proj.weight"].shape[0]
    L = int(z["shapes"].shape[0])
    heads = params["attention_weights.weight"].shape[0] // (L * 4)
    mod = MSDeformAttn(C, L, heads, 4).double()
    mod.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})      # same state-dict keys as the reference
    mod = mod.cuda()
    query = torch.from_numpy(z["query"]).cuda().requires_grad_(True)
    src = torch.from_numpy(z["src"]).cuda().requires_grad_(True)
    out = mod(query, torch.from_numpy(z["reference_points"]).cuda(), src, torch.from_numpy(z["shapes"]).cuda(),
              torch.from_numpy(z["lsi"]).cuda(), torch.from_numpy(z["mask"]).cuda())
    assert rel(out, z["out"]) < 1e-9
    out.backward(torch.from_numpy(z["grad_out"]).cuda())
    assert rel(query.grad, z["grad_query"]) < 1e-9
    assert rel(src.grad, z["grad_src"]) < 1e-9
    for name, p in mod.named_parameters():
        assert rel(p.grad, z["param." + name + ".grad"]) < 1e-9, name


def test_module_float32_tiled_path_close_to_float64():
    """Encoder-shaped fp32 call (takes the LDS-window kernels) against the fp64 golden: fp32 tolerance."""
    z = np.load(os.path.join(GOLDEN, "module_encoder_ref2d.npz"))
    params = {k[len("param."):]: z[k] for k in z.files if k.startswith("param.") and not k.endswith(".grad")}
    mod = MSDeformAttn(params["value_proj.weight"].shape[0], 4, params["attention_weights.weight"].shape[0] // 16, 4)
    mod.load_state_dict({k: torch.from_numpy(v).float() for k, v in params.items()})
    mod = mod.cuda()
    f = lambda k: torch.from_numpy(z[k]).cuda()
    query, src = f("query").float().requires_grad_(True), f("src").float().requires_grad_(True)
    out = mod(query, f("reference_points").float(), src, f("shapes"), f("lsi"), f("mask"))
    assert rel(out, z["out"]) < 1e-4
    out.backward(f("grad_out").float())
    assert rel(query.grad, z["grad_query"]) < 1e-4 and rel(src.grad, z["grad_src"]) < 1e-4
