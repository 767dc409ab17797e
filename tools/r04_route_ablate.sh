#!/bin/bash
# round 4: what bounds the route pass?  rocprofv3 kernel times of rps_route_kernel under the route ablation bits of tile_debug
# (256 no record stores, 512 4-byte records, 1024 runs not announced; the tile kernel is not launched with any of them)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for d in 0 256 512 1024 1280; do
  for loc in init uniform; do
    rm -rf $R/gpurun_out/ra_$d_$loc
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ra_${d}_$loc -- python3 $R/tools/time_calls.py --calls E --loc $loc --bwd 4 --sets 6 --reps 20 --opt tile_debug=$d > /dev/null 2>&1
    python3 - "$R/gpurun_out/ra_${d}_$loc" "$d" "$loc" <<'PY'
import csv, glob, sys, os
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"), key=os.path.getmtime)[-1]
for r in csv.DictReader(open(f)):
    if "rps_" in r["Name"]:
        print(f"dbg={sys.argv[2]:>5s} {sys.argv[3]:8s} {r['Name'][:50]:50s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.1f}")
PY
  done
done
