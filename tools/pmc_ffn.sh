#!/bin/bash
# PMC counters of the fused feed-forward kernel.  Usage: tools/pmc_ffn.sh <tag> "<counters>"
set -e
TAG=$1; CTRS=$2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_ffn_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT -- python3 $ROOT/tools/time_ffn.py --reps 3 > $OUT/probe.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if "ffn_fwd" in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"    {c:32s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
PY
