cd /root/repo
mkdir -p gpurun_out/r4i
rm -f gpurun_out/r4i/status
T="timeout -k 10"
$T 200 python -m pytest tests/test_gpu_sdf.py -q --timeout 150 -k "value_path_default" > gpurun_out/r4i/t.log 2>&1; echo "t rc $?" >> gpurun_out/r4i/status
B="python bench.py --no-cpu-baseline --no-parity-mode --no-pmc --steps 30 --warmup 8"
for i in 1 2; do
  $T 120 $B > gpurun_out/r4i/ab_prod_$i.json 2>/dev/null
  NEUCONW_HIP_LIB=/root/repo/neuralrecon-w_amd/libneuconw_hip_nt.so $T 120 $B > gpurun_out/r4i/ab_nt_$i.json 2>/dev/null
done
NCW_DIST_BACKEND=gloo NCW_BENCH_ONE_GPU_TEST=1 $T 600 python bench.py --gpus 2 --steps 5 --warmup 3 > gpurun_out/r4i/bench_2ranks_one_gpu_gloo.json 2> gpurun_out/r4i/e1; echo "bench2 rc $?" >> gpurun_out/r4i/status
$T 300 python bench.py --bg-eliminate --no-cpu-baseline --no-parity-mode > gpurun_out/r4i/bench_elim.json 2> gpurun_out/r4i/e2; echo "elim rc $?" >> gpurun_out/r4i/status
cat gpurun_out/r4i/status; tail -3 gpurun_out/r4i/t.log
python - <<'P'
import json,glob
for f in sorted(glob.glob('/root/repo/gpurun_out/r4i/ab_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); k=d['roofline']['per_step_kernel_ms']
    print(f.split('/')[-1], round(d['ms_per_step'],4), {x:k[x] for x in ('ncw_wgrad_tiled','ncw_sdf_bwd','ncw_nerf_bwd','ncw_sdf_fwd')})
P
