#!/usr/bin/env python3
"""Which call sites of an eager composed step run which aten ops (TorchDispatchMode + the innermost repo frame; the backward on the
calling thread so that custom Functions' backward frames are seen; built-in backward nodes show up under bench_step's backward call)."""
import collections
import os
import sys

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_step                                            # noqa: E402

dev = torch.device("cuda", 0)
model = bench_step.Step(n_img=2, dev=dev)
model.timing = False
images, mask, targets = model.batch()
model.prepare(mask, targets)
params = [p for p in model.parameters() if p.requires_grad]
SKIP = ("aten.view", "aten.empty", "aten.as_strided", "aten.detach", "aten.slice", "aten.select", "aten.reshape", "aten.t.", "aten.transpose", "aten.unsqueeze",
        "aten.expand", "aten.permute", "aten._unsafe_view", "aten.alias", "aten.squeeze", "aten.lift_fresh", "aten.unbind", "aten.split", "aten.is_", "aten.sym_",
        "aten._local_scalar_dense", "aten.resize_", "aten.set_", "aten.unfold", "aten.narrow", "aten.flatten", "aten.chunk", "aten.contiguous")


class Count(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.acc = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            f = sys._getframe(1)
            site = "?"
            while f is not None:
                fn = f.f_code.co_filename
                if fn.startswith(ROOT) and "/tools/" not in fn:
                    site = f"{os.path.relpath(fn, ROOT)}:{f.f_lineno} {f.f_code.co_name}"
                    break
                f = f.f_back
            self.acc[(name, site)] += 1
        return func(*args, **(kwargs or {}))


def step():
    for p in params:
        p.grad = None
    loss = model(images, mask, targets)
    loss.backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)
with Count() as c:
    step()
torch.cuda.synchronize()
by_site = collections.Counter()
for (name, site), n in c.acc.items():
    by_site[site] += n
print("== ops per call site (top 60)")
for site, n in by_site.most_common(60):
    ops = collections.Counter({name: k for (name, s), k in c.acc.items() if s == site})
    print(f"{n:5d}  {site:70s} " + ", ".join(f"{k.replace('aten.', '')} x{v}" for k, v in ops.most_common(6)))
print("total ops counted:", sum(by_site.values()))
