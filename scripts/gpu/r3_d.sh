#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3d; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1500 python -m pytest tests -m gpu -q -s ) > $O/gpu_tests.log 2>&1
echo "gpu_tests rc=$?" >> $O/summary.txt
grep -E "passed|failed" $O/gpu_tests.log | tail -3; grep -E "^FAILED|Error" $O/gpu_tests.log | head -20
