#!/bin/bash
# secondary bench lines of the final build
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03; mkdir -p $O
timeout 400 python bench.py --bg-eliminate --no-cpu-baseline > $O/bench_elim_v3.json 2>/dev/null
timeout 400 python bench.py --config voxel --no-cpu-baseline --no-parity-mode > $O/bench_voxel_v3.json 2>/dev/null
timeout 400 python bench.py --config grid512 > $O/bench_grid512_v3.json 2>/dev/null
timeout 400 python bench.py --config shipped --no-cpu-baseline > $O/bench_shipped_2048rays_v3.json 2>/dev/null
timeout 400 python bench.py --config shipped --no-cpu-baseline --no-parity-mode --no-pmc --prec f32 --steps 5 --warmup 2 > $O/bench_shipped_2048rays_f32_v3.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*_v3.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print("%-40s %s %.4g %s  %.3f ms"%(f.split("/")[-1], d.get("metric","")[:30], d["value"], d["unit"], d.get("ms_per_step",0)))
    except Exception as e: print(f,"ERR",e)
PY
