#!/bin/bash
# round 4: MFMA-pipe and LDS-pipe busy shares of the MFMA kernels (separate PMC passes, kernel tracing only)
R=${GRAFT_REPO_ROOT:-.}
cd $R
C="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES"
bash tools/pmc_kernel.sh r04m_conv "$C" conv_ time_backbone.py --train --ours-only --reps 2
bash tools/pmc_kernel.sh r04m_dec "$C" _kernel time_decoder_layer.py --reps 2
bash tools/pmc_kernel.sh r04m_lin "$C" lin256 lin256_probe.py
bash tools/pmc_kernel.sh r04m_ffn "$C" ffn_ time_ffn.py --reps 2
