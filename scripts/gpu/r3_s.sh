#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3s; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_sdf.py tests/test_gpu_sdf_train.py tests/test_gpu_fullsize.py tests/test_gpu_repro.py -q -x -s -k "512 or repro" ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/summary.txt
( time timeout 600 python bench.py --config shipped --prec f32 --no-pmc --no-parity-mode --no-cpu-baseline --steps 5 --warmup 2 ) > $O/bench_shipped_f32.json 2> $O/bench.err
grep -E "passed|failed" $O/tests.log | tail -2; grep -E "^E " $O/tests.log | head; grep "W=512" $O/tests.log | grep f32 | head; cat $O/summary.txt
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r3s/bench_shipped_f32.json") if l.startswith("{")][0])
print(d["value"]/1e6, d["ms_per_step"], {k:v for k,v in d["roofline"]["per_step_kernel_ms"].items() if v>0.3}, d["roofline"]["step_frac_of_mfma_peak"])
PY
