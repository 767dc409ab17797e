#!/bin/bash
# Run on the GPU box (through gpurun): bench line + rocprofv3 kernel trace + HBM traffic counters, into gpurun_out/<tag>/.
#   tools/collect_profiles.sh r01
# rocprofv3 is given the program itself after `--` (python3 bench.py ...), never a wrapper; --pmc runs are separate passes
# with kernel tracing only, as the pool requires.
set -e -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 $ROOT/bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json; echo
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-step > $OUT/trace_bench.json 2> $OUT/trace.err
# the headline distribution alone (no sigma4 / uniform / bf16 / mosaic passes): the rows of the routed kernels in THIS trace are what `roofline.avg_launch_us` is compared with
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_init -- python3 $ROOT/bench.py --steps 10 --warmup 3 --loc init --no-bf16 --no-em --no-settle --no-cpu-baseline --no-full-step > $OUT/trace_init_bench.json 2> $OUT/trace_init.err
# (--no-em --no-settle: the persistent kernels have ONE grid size whatever the shape, so the headline shape E and the mosaic shape Em are counted in passes of their own)
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-full-step --no-em --no-settle > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-full-step --no-em --no-settle > $OUT/pmc_write.json 2> $OUT/pmc_write.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_em -- python3 $ROOT/tools/time_calls.py --calls Em --loc init --fwd 2 --bwd 4 --sets 3 --reps 4 > $OUT/pmc_fetch_em.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_em -- python3 $ROOT/tools/time_calls.py --calls Em --loc init --fwd 2 --bwd 4 --sets 3 --reps 4 > $OUT/pmc_write_em.txt 2>&1
# the composed training step (bench_step.py), eager, for profiles/<tag>_step_kernels.md (tools/step_kernels.py <dir> 13 <tag>)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/step_trace -- python3 $ROOT/bench_step.py --no-graph --steps 10 --warmup 3 > $OUT/step_trace.json 2> $OUT/step_trace.err
echo collected $OUT
