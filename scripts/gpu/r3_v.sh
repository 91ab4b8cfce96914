#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3v; mkdir -p $O
export NEUCONW_HIP_LIB=$PWD/neuralrecon-w_amd/libneuconw_hip_w4.so
( timeout 600 python -m pytest tests/test_gpu_sdf.py -q -x -s -k "512 and bf16" ) 2>&1 | grep -E "passed|failed|rel err|^E " | head
for e in 0 8 1 2 3; do NCW_P16_EXP=$e timeout 300 python scripts/time_infer16.py 2>/dev/null | grep "bf16" | sed "s/^/W4 EXP=$e /"; done | tee $O/exp_w4.log
unset NEUCONW_HIP_LIB
for e in 0; do NCW_P16_EXP=$e timeout 300 python scripts/time_infer16.py 2>/dev/null | grep "bf16" | sed "s/^/W8 EXP=$e /"; done | tee $O/exp_w8.log
