"""Which section of the composed step (bench_step.py) cannot be captured into a HIP graph?  Captures the step cut off after each
section in turn (forward + backward of a surrogate loss) and prints OK / the error per section.  GPU box only."""
import contextlib
import os
import sys
import traceback

import torch

sys.path.insert(0, ".")
import bench_step as B

dev = torch.device("cuda", 0)
model = B.Step(n_img=2, dev=dev)
images, mask, targets = model.batch()
model.prepare(mask, targets)
model.timing = False
params = [p for p in model.parameters() if p.requires_grad]
model(images, mask, targets).backward()
for _ in range(int(os.environ.get("PROBE_EAGER", "0"))):
    model.timing = os.environ.get("PROBE_TIMING", "0") == "1"
    for p in params:
        p.grad = None
    model(images, mask, targets).backward()
    if model.timing:
        model.section_ms()
model.timing = False
indices = model.pack_indices(model.last_indices, targets)
torch.cuda.synchronize()


def step():
    for p in params:
        p.grad = None
    model(images, mask, targets, indices).backward()


anomaly = os.environ.get("PROBE_ANOMALY", "1") == "1"
for name in sys.argv[1:] or ["backbone", "input_proj", "encoder", "two_stage", "dn", "decoder", "heads", None]:
    name = None if name == "full" else name
    model.stop_at = name
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            step()
            step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with (torch.autograd.detect_anomaly(check_nan=False) if anomaly else contextlib.nullcontext()), torch.cuda.graph(g, stream=s):
            step()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        print(f"== {name}: OK", flush=True)
    except Exception as e:
        print(f"== {name}: {type(e).__name__} {str(e)[:120]}", flush=True)
        traceback.print_exc()
        break
