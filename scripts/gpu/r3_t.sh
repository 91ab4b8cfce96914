#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3t; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_sdf.py -q -x -s -k "512" ) > $O/tests.log 2>&1
echo "tests rc=$?" > $O/summary.txt
grep -E "passed|failed" $O/tests.log | tail -2; grep -E "^E " $O/tests.log | head; grep "W=512" $O/tests.log | head -12
timeout 300 python scripts/time_infer16.py > $O/t1.log 2>&1; NCW_PP16=0 timeout 300 python scripts/time_infer16.py > $O/t0.log 2>&1
paste -d'\n' $O/t1.log $O/t0.log
