cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 600 python -m pytest tests/test_gpu_layers.py tests/test_abi.py -x -q 2>&1 | tail -3
timeout -k 10 300 python tools/time_decoder_layer.py --only bf16 2>&1 | grep -v amdgpu | tail -5
