#!/usr/bin/env python3
"""Per-(kernel, grid) table of the library's MFMA kernels in a rocprofv3 kernel trace of the composed step (tuning aid).
    python tools/step_shapes.py gpurun_out/r04_step/step_trace 13"""
import collections
import csv
import glob
import os
import re
import sys

f = sorted(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"), key=os.path.getmtime)[-1]
steps = int(sys.argv[2])
keys = sys.argv[3].split(",") if len(sys.argv) > 3 else ("conv_", "lin256", "narrow_linear", "ffn_", "attn_")
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    if not any(k in n for k in keys):
        continue
    gx, gy, gz = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])
    nm = re.sub(r"^void ", "", n).split("(")[0][:34]
    acc[(nm, gx, gy, gz)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = collections.defaultdict(lambda: [0.0, 0])
print(f"{'kernel':34s} {'wg x':>6s} {'y':>5s} {'z':>3s} {'n/step':>7s} {'avg us':>8s} {'ms/step':>8s}")
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    tot[k[0]][0] += sum(v) / steps / 1e3
    tot[k[0]][1] += len(v) / steps
    if sum(v) / steps / 1e3 > 0.05:
        print(f"{k[0]:34s} {k[1]:6d} {k[2]:5d} {k[3]:3d} {len(v)/steps:7.1f} {sum(v)/len(v):8.1f} {sum(v)/steps/1e3:8.3f}")
print()
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:36s} {v[0]:.3f} ms  {v[1]:.0f} launches")
