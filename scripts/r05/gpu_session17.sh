#!/bin/bash
# Round-5 GPU session 17: six more ray batches (bench.py --seed) through the final kernels: the trained-weights parity point of each.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05u; mkdir -p $OUT; rm -f $OUT/status
for S in 5000 6000 7000 8000 9000 10000; do
  timeout -k 10 200 python bench.py --seed $S --no-pmc --no-parity-mode > $OUT/bench_seed$S.json 2>/dev/null; echo "seed $S rc $?" >> $OUT/status
done
cat $OUT/status
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05u/bench_seed*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); p=d['parity']; t=p['trained_40_steps_inv_s_403']; f32=p['f32_mode_trained_40_steps_inv_s_403']
    print(f.split('seed')[1][:-5], round(d['ms_per_step'],3), 'init %.2e/%.2e/%.2e'%(p['colour'],p['depth'],p['weights_sum']), 'trained colour %.2e p99 %.2e above %.4f depth %.2e ws %.2e | f32 mode colour %.2e depth %.2e | fixed_z colour %.2e'%(t['colour'],t['colour_p99'],t['colour_rays_above_1e-4'],t['depth'],t['weights_sum'],f32['colour'],f32['depth'],t['fixed_z']['colour']))
P
