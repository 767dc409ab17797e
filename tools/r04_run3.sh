#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py tests/test_gpu_runtime.py -x -q -m gpu 2>&1 | tail -3
bash tools/rps_trace.sh init
bash tools/rps_trace.sh uniform
python tools/rps_stamps.py --call E --loc init 2>&1 | grep -v amdgpu
