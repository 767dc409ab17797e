cd ${GRAFT_REPO_ROOT:-.}
for f in "" "--no-overlap"; do echo "== $f"; timeout -k 10 600 python bench_step.py $f > gpurun_out/full_step.log 2>&1; grep -v Warning gpurun_out/full_step.log | grep "\"ms\"\|img_per_s\|teacher\|Error\|error" | head -12; done
