#!/bin/bash
# Tuning aid: per-kernel times of the routed backward (rocprofv3 kernel trace) for one distribution.  Extra args: --opt k=v ...
LOC=${1:-init}; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/rps_trace_$LOC -- python3 $GRAFT_REPO_ROOT/tools/time_calls.py --calls E --loc $LOC --bwd 4 "$@" > /dev/null 2>&1
python3 - "$GRAFT_REPO_ROOT/gpurun_out/rps_trace_$LOC" <<'PY'
import csv, glob, sys, os
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"), key=os.path.getmtime)[-1]
for r in csv.DictReader(open(f)):
    if "rps_" in r["Name"]:
        print(f"{r['Name'][:60]:60s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.1f}")
PY
