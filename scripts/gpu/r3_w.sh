#!/bin/bash
# A/B of a tagged library build against the product build on one box: headline + shipped, short runs, alternating
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${TAG:-noieee}
O=gpurun_out/r3w_$TAG; mkdir -p $O
B="--no-pmc --no-parity-mode --no-cpu-baseline --steps 30 --warmup 5"
for rep in 1 2; do
  for lib in base $TAG; do
    if [ $lib = base ]; then unset NEUCONW_HIP_LIB; else export NEUCONW_HIP_LIB=$PWD/neuralrecon-w_amd/libneuconw_hip_$TAG.so; fi
    timeout 300 python bench.py $B > $O/head_${lib}_$rep.json 2>/dev/null
    timeout 300 python bench.py $B --config shipped > $O/ship_${lib}_$rep.json 2>/dev/null
  done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        k=d["roofline"]["per_step_kernel_ms"]
        print(f.split("/")[-1], "%.3f ms"%d["ms_per_step"], {a:round(b,3) for a,b in k.items() if b>0.12})
    except Exception as e: print(f, "ERR", e)
PY
