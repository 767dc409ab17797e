#!/bin/bash
# Per-section kernel tables of the composed step: rocprofv3 kernel stats of the step cut off after each section (gpurun box).
#   gpurun -- bash tools/step_sections.sh r03s      then: python tools/step_sections.py gpurun_out/r03s 7
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out/${1:-sections}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for s in backbone input_proj encoder two_stage dn decoder heads; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$s -- python3 $ROOT/bench_step.py --steps 5 --warmup 2 --stop-at $s > $O/$s.json 2> $O/$s.err || exit 1
  echo done $s
done
