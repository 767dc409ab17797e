#!/usr/bin/env python3
"""Tuning aid: the two-stage selection score kernel (csrc/cls_mfma.hip) against the reference's op sequence as PyTorch fp32 / bf16 ops.

    python tools/time_cls.py [--tokens 44646] [--classes 1204] [--reps 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from richsem_amd.two_stage import ClassScorer   # noqa: E402


def timeit(fn, reps):
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=44646)
    ap.add_argument("--classes", type=int, default=1204)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    torch.manual_seed(0)
    T, C, P = args.tokens, args.classes, 1024
    mem = torch.randn(T, 256, device="cuda")
    wp = torch.randn(P, 256, device="cuda") * P ** -0.5
    text = torch.randn(C, P, device="cuda")
    ls = torch.tensor(2.659)

    def ref(x, w, t):
        f = x @ w.t()
        f = f / f.norm(dim=-1, keepdim=True)
        tt = t / t.norm(dim=-1, keepdim=True)
        return (ls.exp().to(x.dtype) * (f @ tt.t())).max(-1)[0]

    want = ref(mem.double(), wp.double(), text.double())
    flop_ref = 2.0 * T * 256 * P + 2.0 * T * P * C
    for parts in (2, 1):
        sc = ClassScorer(parts).prepare(wp, text, ls)
        for name, x in (("fp32 memory", mem), ("bf16 memory", mem.to(torch.bfloat16))):
            got = sc.max_logits(x)
            err = (got.double() - want).abs().max().item() / want.abs().max().item()
            t = timeit(lambda: sc.max_logits(x), args.reps)
            n_mfma = (3 if x.dtype == torch.float32 else 2) if parts == 2 else (2 if x.dtype == torch.float32 else 1)
            flop = 2.0 * T * 256 * (16 * ((C + 15) // 16) + 256) * n_mfma
            print(f"parts {parts} {name}: {t:7.1f} us  rel err {err:.1e}  {flop / t / 1e6:6.1f} TFLOP/s issued "
                  f"({flop / t / 1e6 / 2500:.3f} of 2.5 PFLOP/s), {flop_ref / t / 1e6:6.1f} TFLOP/s of the reference's products")
    t32 = timeit(lambda: ref(mem, wp, text), args.reps)
    e32 = (ref(mem, wp, text).double() - want).abs().max().item() / want.abs().max().item()
    m16, w16, t16 = mem.to(torch.bfloat16), wp.to(torch.bfloat16), text.to(torch.bfloat16)
    tb = timeit(lambda: ref(m16, w16, t16), args.reps)
    eb = (ref(m16, w16, t16).double() - want).abs().max().item() / want.abs().max().item()
    print(f"PyTorch fp32 ops: {t32:7.1f} us  rel err {e32:.1e};   PyTorch bf16 ops: {tb:7.1f} us  rel err {eb:.1e}")
    sc = ClassScorer(2).prepare(wp, text, ls)
    memory = mem[: (T // 2) * 2].view(2, -1, 256)
    tk = timeit(lambda: sc.topk_proposals(memory, 900), args.reps)
    tr = timeit(lambda: torch.topk(ref(memory, wp, text), 900, dim=1)[1], args.reps)
    print(f"scores + top-900: {tk:7.1f} us   (PyTorch fp32 ops + torch.topk: {tr:7.1f} us)")


if __name__ == "__main__":
    main()
