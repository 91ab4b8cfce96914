#!/bin/bash
# Round-5 GPU session 14: SQ counters of nerf_refineS_kernel (what bounds its 0.09 ms: 26 us of MFMAs, 20 us of LDS reads on paper).
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05r; mkdir -p $OUT
timeout -k 10 600 python scripts/pmc_pass.py $OUT/pmc_refine.json "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC" "GRBM_GUI_ACTIVE GRBM_COUNT" > $OUT/pmc_refine.log 2>&1
grep -E "refine|nerf_fwdB|sdf_inferS2" $OUT/pmc_refine.log | cut -c1-900
