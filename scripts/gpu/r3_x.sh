#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3x; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/gpu_tests.log 2>&1
echo "gpu_tests rc=$?" > $O/summary.txt
( python __graft_entry__.py --smoke ) > $O/smoke.log 2>&1
echo "smoke rc=$?" >> $O/summary.txt
timeout 900 bash scripts/collect_profiles.sh v3 > $O/collect.log 2>&1
echo "collect rc=$?" >> $O/summary.txt
tail -3 $O/gpu_tests.log; cat $O/summary.txt; grep smoke $O/smoke.log; tail -c 300 gpurun_out/r03/bench_v3.json
