#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3c; mkdir -p $O
( timeout 300 python scripts/diag/split_check.py ) > $O/split_check.log 2>&1
echo "split_check rc=$?" >> $O/summary.txt
( timeout 600 python scripts/diag/trained_point_parity.py --precs f16 --json $O/trained_point_f16split.json ) > $O/trained_point.log 2>&1
echo "trained rc=$?" >> $O/summary.txt
( timeout 600 python bench.py --no-pmc --no-parity-mode ) > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" >> $O/summary.txt
cat $O/split_check.log | tail -8; tail -12 $O/trained_point.log; cat $O/summary.txt
