#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3m; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_sdf.py tests/test_gpu_sdf_train.py tests/test_gpu_fullsize.py -q -x -s ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/summary.txt
( time timeout 600 python bench.py --config shipped --no-pmc --no-parity-mode --no-cpu-baseline ) > $O/bench_shipped2048.json 2> $O/bench_shipped.err
( NEUCONW_SDF_SPLIT=0 timeout 600 python bench.py --config shipped --no-pmc --no-parity-mode --no-cpu-baseline ) > $O/bench_shipped2048_plain.json 2>> $O/bench_shipped.err
grep -E "passed|failed" $O/tests.log | tail -2; grep -E "^E " $O/tests.log | head; grep "W=512" $O/tests.log | head -20; cat $O/summary.txt
python - <<'PY'
import json
for f in ("bench_shipped2048","bench_shipped2048_plain"):
    try:
        d=json.loads([l for l in open("gpurun_out/r3m/%s.json"%f) if l.startswith("{")][0])
        print(f, d["value"]/1e6, d["ms_per_step"], {k:v for k,v in d["roofline"]["per_step_kernel_ms"].items() if v>0.1})
    except Exception as e: print(f, "ERR", e)
PY
