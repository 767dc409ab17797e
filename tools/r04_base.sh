#!/bin/bash
# round-4 baseline: the routed backward per level (tile_debug bits 4..6 = 1 + level: only that level's tiles work)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
{
for d in 0 16 32 48 64; do
  echo "== tile_debug=$d"
  timeout -k 10 200 python tools/time_calls.py --calls E --loc init --bwd 4 --sets 6 --reps 20 --opt tile_debug=$d 2>&1 | grep "bwd"
  timeout -k 10 200 python tools/rps_stamps.py --call E --loc init --opt tile_debug=$d 2>&1 | grep -v amdgpu
done
echo "== uniform"
timeout -k 10 200 python tools/time_calls.py --calls E --loc uniform,sigma4 --bwd 4 --sets 6 --reps 20 2>&1 | grep "bwd"
} > gpurun_out/r04_base.txt 2>&1
