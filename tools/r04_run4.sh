#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
for w in 2 3 4 6; do echo "route_wgs=$w"; bash tools/rps_trace.sh init --opt rps_route_wgs=$w | grep route; done
