cd /root/repo
mkdir -p gpurun_out/r4k
T="timeout -k 10"
B="python bench.py --no-cpu-baseline --no-parity-mode --no-pmc --steps 30 --warmup 8"
for i in 1 2; do
  $T 120 $B > gpurun_out/r4k/ab_prod_$i.json 2>/dev/null
  for v in nta ntb ntc ntd; do
    NEUCONW_HIP_LIB=/root/repo/neuralrecon-w_amd/libneuconw_hip_$v.so $T 120 $B > gpurun_out/r4k/ab_${v}_$i.json 2>/dev/null
  done
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('/root/repo/gpurun_out/r4k/ab_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); k=d['roofline']['per_step_kernel_ms']
        print(f.split('/')[-1], round(d['ms_per_step'],4), d['config']['final_loss'], {x:k[x] for x in ('ncw_wgrad_tiled','ncw_sdf_bwd','ncw_nerf_bwd','ncw_sdf_fwd','ncw_nerf_fwd','ncw_color_bwd','ncw_color_fwd')})
    except Exception as e:
        print(f, 'FAILED', e)
P
