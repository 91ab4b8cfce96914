#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r3g; mkdir -p $O
export TMPDIR=/tmp
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS"
G2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
( cd /tmp && timeout 600 python $OLDPWD/scripts/pmc_pass.py $O/pmc_v2.json "$G1" "$G2" ) > $O/pmc_v2.log 2>&1
( cd /tmp && NCW_SPLIT_V=1 timeout 600 python $OLDPWD/scripts/pmc_pass.py $O/pmc_v1.json "$G1" "$G2" ) > $O/pmc_v1.log 2>&1
grep -E "sdf_inferS|sdf_fwdS|sdf_inferC|sdf_fwdB" $O/pmc_v2.log $O/pmc_v1.log
