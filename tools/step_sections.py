#!/usr/bin/env python3
"""Differences between the kernel stats of the composed step cut off after consecutive sections (tools/step_sections.sh): what each
section launches, forward + backward, per step.   python tools/step_sections.py gpurun_out/r03s 7 [full_step_trace_dir]"""
import csv
import glob
import os
import re
import sys

SECTIONS = ["backbone", "input_proj", "encoder", "two_stage", "dn", "decoder", "heads"]


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"at::native::(vectorized_elementwise_kernel|elementwise_kernel_manual_unroll|elementwise_kernel|unrolled_elementwise_kernel)<[^,]*, ", "ew<", n)
    return n[:100]


def load(d, steps):
    f = glob.glob(os.path.join(d, "*", "*_kernel_stats.csv"))[0]
    return {r["Name"]: (int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / 1e6 / steps) for r in csv.DictReader(open(f))}


def main():
    d, steps = sys.argv[1], int(sys.argv[2])
    stages = [(s, load(os.path.join(d, s), steps)) for s in SECTIONS if os.path.isdir(os.path.join(d, s))]
    if len(sys.argv) > 3:
        stages.append(("teacher+criterion", load(sys.argv[3], int(sys.argv[4]))))
    prev = {}
    for name, cur in stages:
        diff = {k: (cur[k][0] - prev.get(k, (0, 0))[0], cur[k][1] - prev.get(k, (0, 0))[1]) for k in cur}
        tot_c, tot_t = sum(v[0] for v in diff.values()), sum(v[1] for v in diff.values())
        lib = sum(v[1] for k, v in diff.items() if "at::native" not in k and "rocprim" not in k and "__amd_rocclr" not in k and not k.startswith("Cijk"))
        print(f"\n== {name}: {tot_t:.2f} ms, {tot_c:.0f} launches per step (library kernels {lib:.2f} ms)")
        for k, (c, t) in sorted(diff.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get('TOP', '14'))]:
            if c > 0.01 or t > 0.001:
                print(f"   {t:7.3f} ms  {c:6.1f} x  {short(k)}")
        prev = cur


if __name__ == "__main__":
    main()
