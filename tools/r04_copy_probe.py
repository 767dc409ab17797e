import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch.profiler import profile, ProfilerActivity
from richsem_amd.conv import ConvBNAct
from richsem_amd.backbone import Bottleneck
m = Bottleneck(512, 128, stride=1, downsample=False).cuda()
x = torch.randn(2, 50, 84, 512, device="cuda").to(torch.bfloat16).requires_grad_(True)
for _ in range(2):
    for p in m.parameters(): p.grad = None
    m(x).float().sum().backward()
torch.cuda.synchronize()
for p in m.parameters(): p.grad = None
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    y = m(x)
    y.float().sum().backward()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cpu_time_total", row_limit=40, max_name_column_width=60))
