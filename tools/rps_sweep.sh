#!/bin/bash
# Tuning aid: routed backward against tile size, slab split (max chunks per workgroup), route grid and walk-unit length (one box: A/B valid).
cd ${GRAFT_REPO_ROOT:-.}
run() { echo "== $*"; timeout -k 10 100 python tools/time_calls.py --calls E --loc init,sigma4,uniform --bwd 4 --sets 6 --reps 20 "$@" 2>&1 | grep bwd; }
run
for t in 12 14; do run --opt rps_tile=$t; done
for c in 4 6 8 16 24; do run --opt rps_max_chunks=$c; done
for w in 2 8; do run --opt rps_route_wgs=$w; done
for s in 3 5; do run --opt rps_seg_shift=$s; done
run
