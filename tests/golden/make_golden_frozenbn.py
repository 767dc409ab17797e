#!/usr/bin/env python3
"""Golden vector for the frozen BatchNorm fold of SURVEY.md section 8a row a10, generated from the REFERENCE's own ``FrozenBatchNorm2d``.

Run in the build container only:  python tests/golden/make_golden_frozenbn.py

``models/richsem/backbone.py`` imports torchvision at module level (absent from the image), so the class ``FrozenBatchNorm2d``
(backbone.py:18-56) is cut out of the source with ``ast`` and executed on its own; its ``forward`` is what produces ``y``.
"""
import ast
import os

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

path = f"{REF}/models/richsem/backbone.py"
tree = ast.parse(open(path).read())
body = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "FrozenBatchNorm2d"]
assert len(body) == 1
ns = {"torch": torch}
exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
g = torch.Generator().manual_seed(21)
bn = ns["FrozenBatchNorm2d"](24)
bn.weight.copy_(1 + 0.3 * torch.randn(24, generator=g))
bn.bias.copy_(0.2 * torch.randn(24, generator=g))
bn.running_mean.copy_(0.5 * torch.randn(24, generator=g))
bn.running_var.copy_(torch.rand(24, generator=g) * 2 + 1e-3)
x = torch.randn(2, 24, 5, 7, generator=g)
y = bn(x)
np.savez_compressed(os.path.join(OUT, "frozenbn_fold.npz"), x=x.numpy(), y=y.numpy(), **{k: v.numpy() for k, v in bn.state_dict().items()})
print("frozenbn_fold", tuple(y.shape))
