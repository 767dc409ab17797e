#!/usr/bin/env python3
"""Generate tests/golden/clip_resnet_*.npz from the REFERENCE's own ModifiedResNet (clip/model.py:94-167, pure torch).

Run in the build container only (needs /root/reference):  python tests/golden/make_golden_clip_resnet.py

No CLIP checkpoint is available offline, so the reference class is instantiated at reduced width / depth and its parameters AND
BatchNorm running statistics are filled from a numpy generator (tests/clip_resnet_params.py: the same function rebuilds them in the
tests, so the fixture holds only the input image batch and the reference's fp32 outputs -- the stride-32 feature map of
``forward(x, ret_sp=True)`` and the pooled embedding of ``forward(x)``)."""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from clip_resnet_params import CASES, fill_state_dict      # noqa: E402

REF = "/root/reference/clip/model.py"


def main():
    spec = importlib.util.spec_from_file_location("_ref_clip_model", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for name, (layers, width, heads, out_dim, res, shape, seed) in CASES.items():
        m = mod.ModifiedResNet(layers, out_dim, heads, input_resolution=res, width=width).eval()
        m.load_state_dict(fill_state_dict(m.state_dict(), seed))
        rng = np.random.default_rng(seed + 1)
        x = torch.from_numpy(rng.normal(0, 1, shape).astype(np.float32))
        with torch.no_grad():
            _, fmap = m(x, ret_sp=True)
            arrays = {"x": x.numpy(), "fmap": fmap.numpy()}
            if shape[2] == res and shape[3] == res:
                arrays["embed"] = m(x).numpy()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
        print(name, tuple(x.shape), "->", tuple(fmap.shape), "embed" in arrays)


if __name__ == "__main__":
    main()
