#!/bin/bash
# One gpurun call: the whole GPU test suite, then timings of the operator's calls.  Usage: gpurun -- bash tools/gpu_check.sh <tag>
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/${1:-check}; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; tail -5 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/time_calls.py --calls E,Dd --loc init,sigma4,uniform --fwd 1,2 --bwd 1,4 --sets 6 --reps 30 2>&1 | grep -v amdgpu.ids | tee $O/time.log
