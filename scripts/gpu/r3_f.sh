#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3f; mkdir -p $O
( NCW_SPLIT_V=1 timeout 300 python scripts/diag/split_check.py ) > $O/split_check_v1.log 2>&1
( timeout 300 python scripts/diag/split_check.py ) > $O/split_check_v2.log 2>&1
grep "f16" $O/split_check_v1.log $O/split_check_v2.log
