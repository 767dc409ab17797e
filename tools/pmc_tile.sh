#!/bin/bash
# PMC counters of the routed backward's kernels (separate passes), call E, init pattern
R=${GRAFT_REPO_ROOT:-.}
cd $R
rocprofv3 --list-avail 2>/dev/null | grep -oE "SQ_[A-Z0-9_]+" | sort -u > gpurun_out/r05_sq_counters.txt
wc -l gpurun_out/r05_sq_counters.txt
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_LDS_MEM_VIOLATIONS SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  bash tools/pmc_kernel.sh r05t$i "$set" rps_ time_calls.py --calls E --loc init --bwd 4 --sets 6 --reps 10
done
