#!/usr/bin/env python3
"""Diagnostic: where does torch's "AccumulateGrad node's stream does not match" warning fire in bench_step's harnesses?  The warning is
turned into an exception, so the traceback names the backward call; run with `eager` or `graphed`."""
import os
import sys
import traceback
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_step      # noqa: E402

warnings.filterwarnings("error", message=".*AccumulateGrad.*")
kw = dict(height=256, width=320, boxes_per_image=5, seed=0)
which = sys.argv[1] if len(sys.argv) > 1 else "graphed"
try:
    if which == "graphed":
        print(bench_step.run_graphed(2, torch.device("cuda", 0), steps=2, warmup=1, optimizer=False, noise_seed=3, **kw)["loss"])
    else:
        model = bench_step.Step(n_img=2, dev=torch.device("cuda", 0), **kw)
        model.timing = False
        images, mask, targets = model.batch()
        model.prepare(mask, targets)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            pinned = bench_step.pin_grad_accumulators(model.parameters())
            for _ in range(3):
                model(images, mask, targets).backward()
        torch.cuda.synchronize()
        print("eager ok")
    print("no warning")
except Exception:      # noqa: BLE001
    traceback.print_exc()
