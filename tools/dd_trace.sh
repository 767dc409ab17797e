#!/bin/bash
# Tuning aid: per-kernel times (rocprofv3) of the decoder-shaped calls Dd, forward and backward.  tools/dd_trace.sh <tag>
cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/dd
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/dd/$1 -- python3 $GRAFT_REPO_ROOT/tools/time_calls.py --calls Dd --loc init,uniform --fwd 0 --bwd 1 --sets 6 --reps 30 > $GRAFT_REPO_ROOT/gpurun_out/dd/$1.txt 2>&1
grep "^Dd" $GRAFT_REPO_ROOT/gpurun_out/dd/$1.txt
python3 - $1 <<PY
import csv, glob, os, sys
f = sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/dd/%s/*/*kernel_stats.csv" % sys.argv[1]), key=os.path.getmtime)[-1]
for r in csv.DictReader(open(f)):
    if "msda" in r["Name"]:
        print("  ", r["Name"][:60], r["Calls"], round(float(r["AverageNs"])/1e3, 1))
PY
