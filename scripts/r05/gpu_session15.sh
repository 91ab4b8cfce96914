#!/bin/bash
# Round-5 GPU session 15: the merge of primary and outside depths (z_feed) launched on the background stream; the suites that render, the
# headline line three times.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05s; mkdir -p $OUT; rm -f $OUT/status
T="timeout -k 10"
$T 700 python -m pytest tests/test_gpu_bg_select.py tests/test_gpu_render.py tests/test_gpu_render_only.py tests/test_gpu_fullsize.py tests/test_gpu_repro.py tests/test_gpu_trainer.py tests/test_gpu_voxel.py tests/test_gpu_edges.py -m gpu -q --timeout 600 > $OUT/tests.log 2>&1; echo "tests rc $?" >> $OUT/status
for I in 1 2 3; do
  $T 200 python bench.py --no-pmc --no-parity-mode --no-cpu-baseline > $OUT/bench_$I.json 2>/dev/null; echo "bench $I rc $?" >> $OUT/status
done
NEUCONW_BG_STREAM=0 $T 200 python bench.py --no-pmc --no-parity-mode --no-cpu-baseline > $OUT/bench_one_stream.json 2>/dev/null; echo "one stream rc $?" >> $OUT/status
cat $OUT/status; grep -E "passed|failed" $OUT/tests.log | tail -2; grep -E "^FAILED|^ERROR" $OUT/tests.log | head
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05s/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['ms_per_step'],3), d['roofline'].get('sum_kernel_ms_per_step'))
P
