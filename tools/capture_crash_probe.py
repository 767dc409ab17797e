#!/usr/bin/env python3
"""Round-4 investigation of the round-3 note "hipStreamEndCapture crashes when the eager steps' timing events / loss are still alive"
(bench_step.py).  Runs the composed step's eager phase, then captures it into a HIP graph with ONE suspect kept alive at a time, each
variant in a child process of its own (a crash must not take the others down).  Prints one line per variant.

    python tools/capture_crash_probe.py            # parent: runs every variant
    python tools/capture_crash_probe.py --child VARIANT
Variants: keep = which of the eager phase's objects stay referenced across the capture (none | loss | fwd | ab | all),
          stream = the capture stream (side = the warmed side stream, default = torch's own capture stream).
"""
import contextlib
import gc
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VARIANTS = ["all:same", "none:side", "loss:side", "fwd:side", "ab:side", "all:side", "none:default", "all:default", "loss_detached:side", "events_norecord:side"]


def child(variant):
    import torch
    import bench_step
    keep, cap = variant.split(":")
    dev = torch.device("cuda", 0)
    model = bench_step.Step(n_img=2, dev=dev)
    images, mask, targets = model.batch()
    model.prepare(mask, targets)
    params = [p for p in model.parameters() if p.requires_grad]

    def step(indices=None):
        for p in params:
            p.grad = None
        loss = model(images, mask, targets, indices)
        fwd = None
        if model.timing:
            fwd = torch.cuda.Event(enable_timing=True)
            fwd.record()
        loss.backward()
        return loss, fwd

    # "same": the eager phase runs on the stream that is captured later (the fix bench_step.py uses)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    eager_ctx = torch.cuda.stream(s) if cap == "same" else contextlib.nullcontext()
    with eager_ctx:
        step()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if keep != "events_norecord":
            a.record()
        loss, fwd = step()
        if keep != "events_norecord":
            b.record()
    torch.cuda.synchronize()
    indices = model.pack_indices(model.last_indices, targets)
    torch.cuda.synchronize()
    held = {"none": [], "loss": [loss], "fwd": [fwd], "ab": [a, b], "all": [loss, fwd, a, b], "loss_detached": [loss.detach()],
            "events_norecord": [a, b]}[keep]
    del loss, fwd, a, b
    gc.collect()
    model.timing = False
    model._events = []
    with torch.cuda.stream(s):
        for _ in range(2):
            step(indices)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    print(f"[{variant}] capture begins (held: {[type(h).__name__ for h in held]})", flush=True)
    with (torch.cuda.graph(g, stream=s) if cap in ("side", "same") else torch.cuda.graph(g)):
        step(indices)
    print(f"[{variant}] capture ended", flush=True)
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print(f"[{variant}] OK replay {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
        sys.exit(0)
    for v in (sys.argv[1:] or VARIANTS):
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", v], capture_output=True, text=True, timeout=600)
        tail = [ln for ln in (p.stdout + p.stderr).splitlines() if ln.strip()][-12:]
        print(f"== {v}: rc={p.returncode}", flush=True)
        for ln in tail:
            print("   " + ln[:220], flush=True)
