#!/bin/bash
# Tuning aid: routed backward against tile size and slab split.
for t in 12 14 16; do for c in 3 6 12; do echo "rps_tile=$t rps_max_chunks=$c"; timeout -k 10 100 python tools/time_calls.py --calls E --loc init,uniform --bwd 4 --opt rps_tile=$t --opt rps_max_chunks=$c; done; done
