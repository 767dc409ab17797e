cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_backbone.py tests/test_gpu_ffn.py tests/test_gpu_layers.py tests/test_gpu_module.py -x -q 2>&1 | tail -3
timeout -k 10 600 python bench_step.py > gpurun_out/full_step.log 2>&1; grep -v Warning gpurun_out/full_step.log | grep -v "^ \"what" | tail -32
