cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py -x -q 2>&1 | tail -3
timeout -k 10 200 python tools/time_calls.py --calls E --loc init,sigma4,uniform --bwd 4 --sets 6 --reps 30 2>&1 | grep "bwd"
timeout -k 10 200 python tools/rps_stamps.py --call E --loc init 2>&1 | grep -v amdgpu
