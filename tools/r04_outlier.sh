#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2 3 4; do
  for loc in uniform init; do
    bash tools/rps_trace.sh $loc --reps 60 > /dev/null 2>&1
    python3 - $loc <<'PY'
import csv, glob, os, sys
f = sorted(glob.glob(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/rps_trace_%s/*/*kernel_trace.csv" % sys.argv[1]), key=os.path.getmtime)[-1]
rows=[r for r in csv.DictReader(open(f)) if "rps_tile" in r["Kernel_Name"]]
d=[round((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3) for r in rows]
print(sys.argv[1], "n", len(d), "median", sorted(d)[len(d)//2], "max", max(d), "outliers", [x for x in d if x > 2*sorted(d)[len(d)//2]])
PY
  done
done
