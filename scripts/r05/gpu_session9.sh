#!/bin/bash
# Round-5 GPU session 9: the three networks' pack / weight-norm backward as ONE launch each (packing.pack_many / unpack_many):
# the suites that train, and the headline line three times.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05m; mkdir -p $OUT; rm -f $OUT/status
T="timeout -k 10"
$T 700 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_fullsize.py tests/test_gpu_repro.py tests/test_gpu_train_driver.py tests/test_gpu_ddp.py tests/test_gpu_rccl_world1.py tests/test_gpu_render_only.py -m gpu -q --timeout 600 -s > $OUT/tests.log 2>&1; echo "tests rc $?" >> $OUT/status
for I in 1 2 3; do
  $T 200 python bench.py --no-pmc --no-parity-mode --no-cpu-baseline > $OUT/bench_$I.json 2>/dev/null; echo "bench $I rc $?" >> $OUT/status
done
$T 200 python bench.py --no-pmc --no-cpu-baseline --no-parity-mode --bg-eliminate > $OUT/bench_elim.json 2>/dev/null; echo "elim rc $?" >> $OUT/status
cat $OUT/status; grep -E "passed|failed" $OUT/tests.log | tail -2; grep -E "^FAILED|^ERROR" $OUT/tests.log | head
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05m/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']['per_step_kernel_ms']
    print(f, round(d['ms_per_step'],3), {k:r[k] for k in ('ncw_pack_weights','ncw_unpack_grads','ncw_nerf_refine','ncw_wgrad_tiled') if k in r}, d['roofline'].get('sum_kernel_ms_per_step'))
P
