#!/bin/bash
# round 4, lever (c): region-major work-queue order of the routed tile kernel against heaviest-first: time, WRITE_SIZE, FETCH_SIZE
R=${GRAFT_REPO_ROOT:-.}
cd $R
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -1
for o in 0 1 0 1; do
  echo "== rps_order=$o"
  bash tools/rps_trace.sh init --opt rps_order=$o | grep tile
  bash tools/rps_trace.sh uniform --opt rps_order=$o | grep tile
done
for o in 0 1; do
  for c in WRITE_SIZE FETCH_SIZE; do
    echo "== rps_order=$o $c"
    bash tools/pmc_kernel.sh r04o_${o}_$c "$c" rps_tile time_calls.py --calls E --loc init --bwd 4 --sets 6 --reps 5 --opt rps_order=$o | grep -A1 "rps_tile" | tail -1
  done
done
