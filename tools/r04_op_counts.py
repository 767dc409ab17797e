"""aten-level op counts of one eager composed step (torch.profiler): where the small launches come from."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch.profiler import profile, ProfilerActivity
import bench_step
dev = torch.device("cuda", 0)
model = bench_step.Step(n_img=2, dev=dev)
model.timing = False
images, mask, targets = model.batch()
model.prepare(mask, targets)
params = [p for p in model.parameters() if p.requires_grad]
def step():
    for p in params: p.grad = None
    loss = model(images, mask, targets)
    loss.backward()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.count)
print(f"{'op':60s} {'count':>6s} {'cuda total us':>14s}")
for e in rows[:70]:
    print(f"{e.key[:60]:60s} {e.count:6d} {getattr(e, 'device_time_total', getattr(e, 'cuda_time_total', 0)):14.1f}")
