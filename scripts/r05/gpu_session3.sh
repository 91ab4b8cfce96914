#!/bin/bash
# Round-5 GPU session 3: the NB = 2 / AGPR-weights probe of sdf_inferC, the tests that failed on the probe allocation under capture,
# over-plan sweep of the selection share (headline + shipped, elimination on).
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05c; mkdir -p $OUT; rm -f $OUT/status
T="timeout -k 10"
NEUCONW_HIP_LIB=neuralrecon-w_amd/libneuconw_hip_nb2.so $T 300 python scripts/diag/pp_nb2.py > $OUT/pp_nb2.log 2>&1; echo "pp_nb2 rc $?" >> $OUT/status
$T 900 python -m pytest tests -m gpu -q --timeout 500 > $OUT/full.log 2>&1; echo "pytest -m gpu rc $?" >> $OUT/status
for OP in 1.0 1.3 1.6 2.0 2.5; do
  NCW_SEL_OVERPLAN=$OP $T 200 python bench.py --no-pmc --no-cpu-baseline --no-parity-mode --bg-eliminate > $OUT/headline_elim_op$OP.json 2>/dev/null; echo "headline elim op $OP rc $?" >> $OUT/status
  NCW_SEL_OVERPLAN=$OP $T 200 python bench.py --config shipped --no-pmc --no-cpu-baseline --no-parity-mode --bg-eliminate > $OUT/shipped_elim_op$OP.json 2>/dev/null; echo "shipped elim op $OP rc $?" >> $OUT/status
done
for i in 1 2 3; do
  $T 200 python bench.py --config shipped --no-pmc --no-cpu-baseline --no-parity-mode --graph > $OUT/shipped_dense_graph_$i.json 2>/dev/null; echo "shipped dense graph $i rc $?" >> $OUT/status
done
cat $OUT/status; grep -E "passed|failed" $OUT/full.log | tail -2; grep -E "^FAILED|^ERROR" $OUT/full.log | head; cat $OUT/pp_nb2.log
