#!/bin/bash
# Tuning aid: level-sum kernel with stages switched off (tile_debug: 1 no walk, 2 no flush, 4 no zeroing; wrong results).
for d in 0 1 2 4 3 7; do echo "tile_debug=$d"; timeout -k 10 100 python tools/kernel_probe.py --call Dd --loc init --reps 10 --only bwd --set locality_monitor=0 --set tile_debug=$d; done
cd /tmp && export TMPDIR=/tmp
for d in 0 1 3 7; do timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ls_$d -- python3 $GRAFT_REPO_ROOT/tools/kernel_probe.py --call Dd --loc init --reps 10 --only bwd --set locality_monitor=0 --set tile_debug=$d > /dev/null 2>&1; grep -h "levelsum\|bwd_direct" $GRAFT_REPO_ROOT/gpurun_out/ls_$d/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-60,100-; done
