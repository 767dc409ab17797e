"""Tuning aid: GPU durations of the library's bf16 GEMM at ResNet-50's 1 x 1 convolution shapes (run under rocprofv3 --kernel-trace --stats)."""
import torch
for P, K, N in [(8400, 1024, 256), (8400, 256, 1024), (2100, 2048, 512), (2100, 512, 2048), (33600, 512, 128), (33600, 128, 512), (134400, 64, 256), (134400, 256, 64)]:
    x = torch.randn(P, K, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda").bfloat16()
    for _ in range(20):
        torch.mm(x, w.t())
    torch.cuda.synchronize()
