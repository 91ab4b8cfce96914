#!/bin/bash
# Round-5 GPU session 11: the whole -m gpu suite with its printed measurements (tolerances tightened to the north-star bar where the
# round's numerics allow).
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05o; mkdir -p $OUT; rm -f $OUT/status
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 -s > $OUT/tests_printed.log 2>&1; echo "tests rc $?" >> $OUT/status
cat $OUT/status; grep -E "passed|failed" $OUT/tests_printed.log | tail -2; grep -E "^FAILED|^ERROR" $OUT/tests_printed.log | head; grep -E "single ray|508 \+ 4|samples per ray:" $OUT/tests_printed.log | cut -c1-250
