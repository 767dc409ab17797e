cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 300 python -m pytest tests/test_gpu_backbone.py -x -q 2>&1 | tail -3
timeout -k 10 600 python bench_step.py > gpurun_out/full_step.log 2>&1; grep -v Warning gpurun_out/full_step.log | grep -v "^ \"what" | tail -45
