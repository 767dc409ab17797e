#!/bin/bash
# Tuning aid: forward window kernel against window margin and region size (init pattern).
for m in 6 5 4; do for r in 20 24; do for gr in 1 0; do echo "tile_margin=$m tile_region=$r tile_grow=$gr"; timeout -k 10 100 python tools/time_calls.py --calls E --loc init --fwd 2 --bwd "" --opt tile_margin=$m --opt tile_region=$r --opt tile_grow=$gr; done; done; done
