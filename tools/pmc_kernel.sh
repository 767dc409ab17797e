#!/bin/bash
# PMC counters of any of the library's kernels under one of the timing tools.
#   tools/pmc_kernel.sh <tag> "<counters>" <kernel name substring> <tool.py> [tool args...]
# (rocprofv3 gets the program itself after `--`; counters are collected with kernel tracing only, as the pool requires)
set -e
TAG=$1; CTRS=$2; KSUB=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT -- python3 $ROOT/tools/"$@" > $OUT/probe.log 2>&1 || true
python3 - "$OUT" "$KSUB" <<'PY'
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if sys.argv[2] in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:70] + " grid " + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"    {c:32s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
PY
