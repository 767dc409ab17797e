#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PYTHONPATH=$R timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fprep -- python3 $R/tools/r04_fused_prep_prof.py > /dev/null 2>&1
python3 - "$R/gpurun_out/fprep" <<'PY'
import csv, glob, sys, os
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"), key=os.path.getmtime)[-1]
for r in csv.DictReader(open(f)):
    if "prep" in r["Name"] or "fwd_direct" in r["Name"]:
        print(f"{r['Name'][:110]:110s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.1f}")
PY
