#!/bin/bash
# round 3, GPU call B: the whole GPU suite (elimination in every mode, trained-point tests, 2-rank tests), bench f32-mode timing
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3b; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/gpu_tests.log 2>&1
echo "gpu_tests rc=$?" >> $O/summary.txt
( time timeout 600 python bench.py --no-pmc ) > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" >> $O/summary.txt
tail -15 $O/gpu_tests.log; cat $O/summary.txt
