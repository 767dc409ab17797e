#!/usr/bin/env python3
"""Per-kernel table (calls, average, total, share) from a rocprofv3 --kernel-trace CSV directory, PyTorch / library kernels grouped.

    python tools/trace_table.py gpurun_out/prof_x "title" >> profiles/r02_layer_kernels.md
"""
import collections
import csv
import glob
import sys

d, title = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/*/*_kernel_trace.csv")[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].replace("msda::", "")
    if n.startswith("Cijk_"):
        n = "hipBLASLt GEMM (Cijk_...)"
    elif n.startswith("at::native") or n.startswith("__amd_rocclr") or "at::native" in n:
        n = "PyTorch element-wise / copy / reduce kernels"
    acc[n[:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in acc.values())
print(f"## {title}\n")
print("| kernel | launches | avg µs | total ms | share |\n|---|---|---|---|---|")
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) / tot < 0.004:
        continue
    print(f"| `{k}` | {len(v)} | {sum(v) / len(v):.1f} | {sum(v) / 1e3:.2f} | {100 * sum(v) / tot:.1f} % |")
print()
