#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
bash tools/rps_trace.sh init
bash tools/rps_trace.sh uniform
timeout -k 10 900 python tools/capture_crash_probe.py
