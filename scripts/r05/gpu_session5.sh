#!/bin/bash
# Round-5 GPU session 5: the W = 512 adjoint sweep with hi + lo weights (sdf_fwdS16<., true>): full suite, shipped-shape bench
# with its parity object (dense + elimination), beside NEUCONW_SDF_ADJ_SPLIT=0.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05e; mkdir -p $OUT; rm -f $OUT/status
T="timeout -k 10"
$T 900 python -m pytest tests -m gpu -q --timeout 500 > $OUT/full.log 2>&1; echo "pytest -m gpu rc $?" >> $OUT/status
$T 400 python bench.py --config shipped --no-pmc > $OUT/bench_shipped.json 2> $OUT/bench_shipped.err; echo "shipped rc $?" >> $OUT/status
NEUCONW_SDF_ADJ_SPLIT=0 $T 300 python bench.py --config shipped --no-pmc --no-cpu-baseline --no-parity-mode > $OUT/bench_shipped_adj_off.json 2>/dev/null; echo "shipped adj off rc $?" >> $OUT/status
$T 300 python bench.py --config shipped --no-pmc --no-cpu-baseline --no-parity-mode --bg-eliminate > $OUT/bench_shipped_elim.json 2>/dev/null; echo "shipped elim rc $?" >> $OUT/status
$T 300 python bench.py --config shipped --no-pmc --no-cpu-baseline --no-parity-mode > $OUT/bench_shipped_2.json 2>/dev/null; echo "shipped 2 rc $?" >> $OUT/status
cat $OUT/status; grep -E "passed|failed" $OUT/full.log | tail -2; grep -E "^FAILED|^ERROR" $OUT/full.log | head
