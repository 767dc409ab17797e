#!/bin/bash
# PMC counters of the routed backward's kernels on the vector-memory path (TA / TCP = L1 / TCC = L2), separate passes; call E, init pattern
R=${GRAFT_REPO_ROOT:-.}
cd $R
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(TA|TCP|TCC|TD)_[A-Z0-9_]+" | sort -u > gpurun_out/r05_mem_counters.txt
wc -l gpurun_out/r05_mem_counters.txt
i=0
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  bash tools/pmc_kernel.sh r05m$i "$set" rps_ time_calls.py --calls E --loc init --bwd 4 --sets 6 --reps 6
done
