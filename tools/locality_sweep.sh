#!/bin/bash
# Window ("tiled", v2) vs direct (v1) kernels over the spread of the sampling pattern (sigma in pixels of the sampled
# level), with the share of points that miss the forward windows: the data behind the thresholds of the locality
# monitor.  Third line per setting: automatic mode (what the monitor ends up choosing).
set -e
for j in ${JITTERS:-1 2 3 4 6 8 12 20}; do
  echo "== jitter $j"
  python tools/kernel_probe.py --call E --loc init --jitter $j --stats --reps 12 --set fwd_variant=2,1,0 --set bwd_variant=2,1,0 2>&1 | grep "fwd_variant=\([0-2]\) bwd_variant=\1\|share"
done
echo "== uniform"
python tools/kernel_probe.py --call E --loc uniform --stats --reps 12 --set fwd_variant=2,1,0 --set bwd_variant=2,1,0 2>&1 | grep "fwd_variant=\([0-2]\) bwd_variant=\1\|share"
