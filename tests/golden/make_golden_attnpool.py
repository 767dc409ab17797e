#!/usr/bin/env python3
"""Generate tests/golden/attnpool_*.npz from the REFERENCE's own AttentionPool2d (clip/model.py:58-91).

Run in the build container only (needs /root/reference):  python tests/golden/make_golden_attnpool.py

clip/model.py imports nothing but torch and numpy (SURVEY.md section 8c lists it as importable), so it is loaded by file path and
its class is instantiated with seeded random weights (no CLIP checkpoint is available offline) in fp64 and fp32.  Written: the
module's parameters, an input of ROI features (K, C, H, W) and the module's output -- data only.  Sizes are kept small (the real
attnpool has 4 x 2048 x 2048 weights); the head dimension 64 of CLIP-RN50 is kept in one case."""
import importlib.util
import os

import numpy as np
import torch

REF = "/root/reference/clip/model.py"
OUT = os.path.dirname(os.path.abspath(__file__))

CASES = [  # name, spacial_dim, embed_dim, heads, output_dim, K, dtype
    ("attnpool_s7_c128_h2_f64", 7, 128, 2, 64, 5, torch.float64),       # 7 x 7 grid and head_dim 64 as CLIP-RN50
    ("attnpool_s7_c128_h2_f32", 7, 128, 2, 64, 5, torch.float32),
    ("attnpool_s3_c96_h8_f64", 3, 96, 8, None, 3, torch.float64),        # output_dim None -> embed_dim
    ("attnpool_s9_c64_h4_f64", 9, 64, 4, 40, 2, torch.float64),          # 81 positions: more than one wave of tokens
]


def main():
    spec = importlib.util.spec_from_file_location("_ref_clip_model", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for i, (name, sd, C, H, od, K, dt) in enumerate(CASES):
        torch.manual_seed(100 + i)
        m = mod.AttentionPool2d(sd, C, H, od).to(dt).eval()
        with torch.no_grad():
            for p in m.parameters():          # default init leaves the biases small; make every term count
                p.copy_(torch.randn_like(p) * (0.5 if p.dim() == 1 else p.shape[-1] ** -0.5))
            x = torch.randn(K, C, sd, sd, dtype=dt)
            y = m(x)
        arrays = {"x": x.numpy(), "y": y.numpy(), "num_heads": np.int64(H), "spacial_dim": np.int64(sd),
                  "output_dim": np.int64(od or C)}
        for k, v in m.state_dict().items():
            arrays["param:" + k] = v.numpy()
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
        print(name, tuple(x.shape), "->", tuple(y.shape))


if __name__ == "__main__":
    main()
