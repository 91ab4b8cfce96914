#!/bin/bash
# stash-elimination table: per-kernel times of the product library vs builds without stash stores / loads (timing only)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3z; mkdir -p $O
B="--no-pmc --no-parity-mode --no-cpu-baseline --steps 20 --warmup 5"
for lib in base nostore noload nostash base; do
  if [ $lib = base ]; then unset NEUCONW_HIP_LIB; else export NEUCONW_HIP_LIB=$PWD/neuralrecon-w_amd/libneuconw_hip_$lib.so; fi
  for cfg in "" "--config shipped"; do
    timeout 300 python bench.py $B $cfg > $O/${lib}_$(echo $cfg | tr -d ' -').json 2>/dev/null
  done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        k=d["roofline"]["per_step_kernel_ms"]
        print("%-28s %.3f ms"%(f.split("/")[-1], d["ms_per_step"]), {a.replace("ncw_",""):round(b,3) for a,b in k.items() if b>0.12})
    except Exception as e: print(f, "ERR", e)
PY
