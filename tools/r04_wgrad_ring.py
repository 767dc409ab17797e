#!/usr/bin/env python3
"""A/B of the weight-gradient kernels' operand staging (msda_conv_set_wgrad_ring): the transformer's linear layers at the encoder's 44646 and
the decoder's 2184 tokens, a bottleneck block's grouped launch per ResNet-50 stage, and ResNet-50 forward + backward."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from richsem_amd import _lib                                                     # noqa: E402
from richsem_amd.conv import conv_wgrad_group                                    # noqa: E402
from richsem_amd.functions.linear import linear_wgrad_bf16                       # noqa: E402


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    L = _lib.load()
    torch.manual_seed(0)
    cases = []
    for T in (44646, 2184):
        for cin, cout in ((256, 2048), (2048, 256), (256, 256), (256, 384), (256, 512)):
            x = torch.randn(T, cin, device="cuda").to(torch.bfloat16)
            dy = torch.randn(T, cout, device="cuda").to(torch.bfloat16)
            cases.append((f"linear {T:6d} tokens {cin:4d} -> {cout:4d} (+ bias)", lambda x=x, dy=dy: linear_wgrad_bf16(dy, x, with_bias=True)))
    for name, px, planes, inpl, stride in (("layer2 block", (2, 100, 168), 128, 512, 1), ("layer3 block", (2, 50, 84), 256, 1024, 1),
                                           ("layer4 block", (2, 25, 42), 512, 2048, 1)):
        N, H, W = px
        xin = torch.randn(N, H, W, inpl, device="cuda").to(torch.bfloat16)
        o1 = torch.randn(N, H, W, planes, device="cuda").to(torch.bfloat16)
        dz1, dz2 = torch.randn_like(o1), torch.randn_like(o1)
        dz3 = torch.randn(N, H, W, 4 * planes, device="cuda").to(torch.bfloat16)
        probs = [(dz1, xin, planes, 1, 1, 1, 0, None), (dz2, o1, planes, 3, 3, 1, 1, None), (dz3, o1, 4 * planes, 1, 1, 1, 0, None)]
        cases.append((f"{name}: three weight gradients in one launch", lambda probs=probs: conv_wgrad_group(probs)))
    print(f"{'case':62s} {'register-staged':>16s} {'LDS ring':>10s}")
    for name, fn in cases:
        t = []
        for ring in (0, 1, 0, 1):
            _lib.check(L.msda_conv_set_wgrad_ring(ring))
            t.append(timeit(fn))
        print(f"{name:62s} {min(t[0], t[2]):13.1f} us {min(t[1], t[3]):7.1f} us", flush=True)
    _lib.check(L.msda_conv_set_wgrad_ring(1))


if __name__ == "__main__":
    main()
