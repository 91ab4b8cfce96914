#!/bin/bash
# Round-5 GPU session 2: adjoint sweep with hi + lo weights (sdf_fwdSA), adaptive selection share in the weight-gradient plan,
# the tests that failed in session 1, shipped-shape sweep of the split-K target.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05b; mkdir -p $OUT; rm -f $OUT/status
T="timeout -k 10"
$T 600 python -m pytest tests/test_gpu_render_only.py tests/test_gpu_voxel.py tests/test_gpu_fullsize.py tests/test_gpu_sdf.py tests/test_gpu_sdf_train.py tests/test_gpu_bg_select.py tests/test_gpu_trainer.py -m gpu -q --timeout 500 -s > $OUT/tests.log 2>&1; echo "tests rc $?" >> $OUT/status
$T 400 python bench.py --no-pmc > $OUT/bench_v2.json 2> $OUT/bench_v2.err; echo "bench rc $?" >> $OUT/status
NEUCONW_SDF_ADJ_SPLIT=0 $T 300 python bench.py --no-pmc --no-cpu-baseline --no-parity-mode > $OUT/bench_v2_adj_off.json 2>/dev/null; echo "bench adj off rc $?" >> $OUT/status
for i in 1 2; do
  $T 200 python bench.py --config shipped --no-pmc --no-cpu-baseline --no-parity-mode > $OUT/shipped_dense_$i.json 2>/dev/null; echo "shipped dense $i rc $?" >> $OUT/status
  $T 200 python bench.py --config shipped --no-pmc --no-cpu-baseline --no-parity-mode --bg-eliminate > $OUT/shipped_elim_$i.json 2>/dev/null; echo "shipped elim $i rc $?" >> $OUT/status
done
for W in 256 384 512 640 1024; do
  NCW_WGRAD_TARGET_WGS=$W $T 200 python bench.py --config shipped --no-pmc --no-cpu-baseline --no-parity-mode --bg-eliminate > $OUT/shipped_elim_wgs$W.json 2>/dev/null; echo "shipped elim wgs $W rc $?" >> $OUT/status
  NCW_WGRAD_TARGET_WGS=$W $T 200 python bench.py --config shipped --no-pmc --no-cpu-baseline --no-parity-mode > $OUT/shipped_dense_wgs$W.json 2>/dev/null; echo "shipped dense wgs $W rc $?" >> $OUT/status
done
$T 200 python bench.py --no-pmc --no-cpu-baseline --no-parity-mode --bg-eliminate > $OUT/headline_elim.json 2>/dev/null; echo "headline elim rc $?" >> $OUT/status
cat $OUT/status; grep -E "passed|failed" $OUT/tests.log | tail -2; grep -E "^FAILED|^ERROR" $OUT/tests.log | head
