#!/bin/bash
# eager composed step under rocprofv3 -> gpurun_out/r04_step/step_trace + per-(kernel, grid) table
TAG=${1:-r04_step}
R=${GRAFT_REPO_ROOT:-.}
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/step_trace -- python3 $R/bench_step.py --no-graph --steps 10 --warmup 3 > $R/gpurun_out/$TAG/step_trace.json 2> $R/gpurun_out/$TAG/step_trace.err
cat $R/gpurun_out/$TAG/step_trace.json | head -20
python3 - "$R/gpurun_out/$TAG/step_trace" 13 <<'PY'
import csv, glob, sys, collections, os
f = sorted(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"), key=os.path.getmtime)[-1]
steps = int(sys.argv[2])
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if not any(k in n for k in ("conv_", "lin256", "narrow_linear")):
        continue
    key = (n.split("(")[0][:44], r["Grid_Size"], r.get("Workgroup_Size", ""))
    acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("kernel | grid threads | launches/step | avg us | ms/step")
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[0]:44s} {k[1]:>9s} {len(v)/steps:7.1f} {sum(v)/len(v):8.1f} {sum(v)/steps/1e3:8.3f}")
PY
