#!/bin/bash
# Round-5 GPU session 7: the background NeRF's split-precision refinement (ncw_nerf_refine): unit test, the suites it touches, the bench
# line over four batch seeds (parity of the trained point) beside NEUCONW_NERF_REFINE=0.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05k; mkdir -p $OUT; rm -f $OUT/status
T="timeout -k 10"
$T 600 python -m pytest tests/test_gpu_bg_select.py tests/test_gpu_color_nerf.py tests/test_gpu_fullsize.py tests/test_gpu_render_only.py tests/test_gpu_trainer.py tests/test_gpu_voxel.py -m gpu -q --timeout 500 -s > $OUT/tests.log 2>&1; echo "tests rc $?" >> $OUT/status
for S in 1000 2000 3000 4000; do
  $T 200 python bench.py --seed $S --no-pmc --no-parity-mode > $OUT/bench_seed$S.json 2>/dev/null; echo "seed $S rc $?" >> $OUT/status
done
NEUCONW_NERF_REFINE=0 $T 200 python bench.py --no-pmc --no-cpu-baseline --no-parity-mode > $OUT/bench_refine_off.json 2>/dev/null; echo "refine off rc $?" >> $OUT/status
$T 200 python bench.py --no-pmc --no-cpu-baseline --no-parity-mode --bg-eliminate > $OUT/bench_elim.json 2>/dev/null; echo "elim rc $?" >> $OUT/status
cat $OUT/status; grep -E "passed|failed" $OUT/tests.log | tail -2; grep -E "^FAILED|^ERROR" $OUT/tests.log | head; grep -E "background NeRF at the" $OUT/tests.log
