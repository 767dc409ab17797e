#!/bin/bash
# Locality sweep (run through gpurun): call E, init offsets + N(0, sigma) pixels, every forward / backward variant.
#   fwd: 1 = direct, 2 = window (+ per-point fix-up);  bwd: 1 = direct + level-sum, 4 = routed
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for s in 1 2 3 4 6 8 12 20; do
  timeout -k 10 120 python tools/kernel_probe.py --call E --loc init --jitter $s --reps 6 --stats --set locality_monitor=0 --set fwd_variant=1,2 --only fwd 2>&1 | grep -v amdgpu.ids | sed "s/^/sigma=$s /"
  timeout -k 10 120 python tools/kernel_probe.py --call E --loc init --jitter $s --reps 4 --set locality_monitor=0 --set bwd_variant=1,4 --only bwd 2>&1 | grep -v amdgpu.ids | sed "s/^/sigma=$s /"
done
timeout -k 10 120 python tools/kernel_probe.py --call E --loc uniform --reps 6 --stats --set locality_monitor=0 --set fwd_variant=1,2 --only fwd 2>&1 | grep -v amdgpu.ids | sed "s/^/uniform /"
timeout -k 10 120 python tools/kernel_probe.py --call E --loc uniform --reps 4 --set locality_monitor=0 --set bwd_variant=1,4 --only bwd 2>&1 | grep -v amdgpu.ids | sed "s/^/uniform /"
