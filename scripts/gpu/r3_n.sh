#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3n; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 300 python scripts/diag/split_check.py ) > $O/split_check.log 2>&1
( time timeout 600 python bench.py --no-pmc ) > $O/bench.json 2> $O/bench.err
( time timeout 600 python bench.py --config shipped --no-pmc --no-parity-mode --no-cpu-baseline ) > $O/bench_shipped.json 2>> $O/bench.err
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/gpu_tests.log 2>&1
echo "gpu_tests rc=$?" >> $O/summary.txt
grep -E "f16|f32|bf16" $O/split_check.log; grep -E "passed|failed" $O/gpu_tests.log | tail -1; cat $O/summary.txt
python - <<'PY'
import json
for f in ("bench","bench_shipped"):
    try:
        d=json.loads([l for l in open("gpurun_out/r3n/%s.json"%f) if l.startswith("{")][0])
        print(f, round(d["value"]/1e6,2), round(d["ms_per_step"],3), {k:v for k,v in d["roofline"]["per_step_kernel_ms"].items() if v>0.1})
        if f=="bench": print(d["parity"]["colour"], d["parity"]["at_inv_s_403"], "plain", d["plain_f16_mode"]["ms_per_step"], "bf16", d["alt_mode"]["ms_per_step"], "elim", d["bg_elimination"]["ms_per_step"], "f32", d["parity_mode"]["ms_per_step"], d["parity_mode"]["bg_elimination_ms_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
