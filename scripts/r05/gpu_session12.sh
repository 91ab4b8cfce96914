#!/bin/bash
# Round-5 GPU session 12: nerf_refineS with its trunk weights in rotating register buffers loaded a whole layer ahead (QU units per chunk x
# NB buffers): unit test, then the headline line per variant (probe libraries q44 = 4 x 4, q28 = 2 x 8, old = two halves) -- twice, alternating.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05p; mkdir -p $OUT; rm -f $OUT/status
T="timeout -k 10"
$T 400 python -m pytest tests/test_gpu_bg_select.py tests/test_gpu_render_only.py -m gpu -q --timeout 300 -s > $OUT/tests.log 2>&1; echo "tests rc $?" >> $OUT/status
for I in 1 2; do
for V in main q44 q28 old; do
  if [ $V = main ]; then unset NEUCONW_HIP_LIB; else export NEUCONW_HIP_LIB=$PWD/neuralrecon-w_amd/libneuconw_hip_$V.so; fi
  $T 200 python bench.py --no-pmc --no-parity-mode --no-cpu-baseline > $OUT/bench_${V}_$I.json 2>/dev/null; echo "bench $V $I rc $?" >> $OUT/status
done
done
unset NEUCONW_HIP_LIB
cat $OUT/status; grep -E "passed|failed" $OUT/tests.log | tail -2; grep -E "^FAILED|^ERROR" $OUT/tests.log | head; grep "background NeRF at" $OUT/tests.log
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05p/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']['per_step_kernel_ms']
    print(f, round(d['ms_per_step'],3), 'refine', r.get('ncw_nerf_refine'), 'trained colour', d['parity']['trained_40_steps_inv_s_403']['colour'] if d.get('parity') else None)
P
