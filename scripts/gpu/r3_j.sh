#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3j; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 600 python -m pytest tests/test_gpu_mesh.py tests/test_gpu_grid.py tests/test_gpu_octree_refresh.py -q -x ) > $O/mesh_tests.log 2>&1
echo "mesh rc=$?" >> $O/summary.txt
( time timeout 600 python bench.py --config shipped --no-pmc ) > $O/bench_shipped2048.json 2> $O/bench_shipped.err
( time timeout 600 python bench.py --config shipped --rays 1024 --no-pmc --no-parity-mode --no-cpu-baseline ) > $O/bench_shipped1024.json 2>> $O/bench_shipped.err
tail -5 $O/mesh_tests.log; cat $O/summary.txt
python - <<'PY'
import json
for f in ("bench_shipped2048","bench_shipped1024"):
    try:
        d=json.loads([l for l in open("gpurun_out/r3j/%s.json"%f) if l.startswith("{")][0])
        print(f, d["value"]/1e6, d["ms_per_step"], d["config"]["rays_per_gpu"], d["roofline"]["per_step_kernel_ms"])
    except Exception as e: print(f, "ERR", e)
PY
