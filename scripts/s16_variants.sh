for v in "" _nostash _d8 _d2; do for T in 2 4; do
  NEUCONW_HIP_LIB=$PWD/neuralrecon-w_amd/libneuconw_hip$v.so NCW_SDF16_T=$T timeout 200 python bench.py --config shipped --no-pmc --no-parity-mode --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']['per_step_kernel_ms']
print('$v', $T, round(d['ms_per_step'],3), r.get('ncw_sdf_fwd'), r.get('ncw_sdf_bwd'), r.get('ncw_sdf_infer_rays'))"
done; done
