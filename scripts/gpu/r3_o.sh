#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3o; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 900 python -m pytest tests/test_gpu_train_driver.py "tests/test_gpu_fullsize.py::test_train_step_vs_oracle_after_training" -q -x -s ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/summary.txt
grep -E "after 40|passed|failed|^E " $O/tests.log | head -20; cat $O/summary.txt
