#!/bin/bash
# GPU: (a) split-K target of the weight-gradient launch (workgroups per launch; 768 = 3 rounds of 256 CUs), (b) the background
# NeRF on the second stream or not -- both re-measured under the non-temporal stash policy of round 4.  Prints ms per dense step.
cd "$(dirname "$0")/../.."
B="python bench.py --no-cpu-baseline --no-parity-mode --no-pmc --steps 30 --warmup 8"
for rep in 1 2; do
  for wgs in 512 768 1024 1280 1536; do
    NCW_WGRAD_TARGET_WGS=$wgs timeout 120 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['per_step_kernel_ms']
print('target_wgs $wgs rep $rep: step %.4f ms, wgrad %.4f' % (d['ms_per_step'], k['ncw_wgrad_tiled']))"
  done
  NEUCONW_BG_STREAM=0 timeout 120 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one stream rep $rep: step %.4f ms' % d['ms_per_step'])"
done
