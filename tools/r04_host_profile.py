#!/usr/bin/env python3
"""Where the HOST time of an eager composed step goes (the step is launch-bound when run eagerly): cProfile over a few steps."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_step                                            # noqa: E402

dev = torch.device("cuda", 0)
model = bench_step.Step(n_img=2, dev=dev)
model.timing = False
images, mask, targets = model.batch()
model.prepare(mask, targets)
params = [p for p in model.parameters() if p.requires_grad]


def step():
    for p in params:
        p.grad = None
    loss = model(images, mask, targets)
    loss.backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumtime").print_stats(60)
