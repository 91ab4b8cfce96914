#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3r; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( NEUCONW_BG_STREAM=0 timeout 600 python bench.py --no-pmc --no-cpu-baseline ) > $O/bench_one_stream.json 2> $O/bench.err
( timeout 600 python bench.py --no-pmc --no-cpu-baseline ) > $O/bench_two_streams.json 2>> $O/bench.err
( timeout 600 python bench.py --graph --no-pmc --no-cpu-baseline --no-parity-mode ) > $O/bench_two_streams_graph.json 2>> $O/bench.err
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/gpu_tests.log 2>&1
echo "gpu_tests rc=$?" >> $O/summary.txt
grep -E "passed|failed" $O/gpu_tests.log | tail -1; grep -E "^E " $O/gpu_tests.log | head -5; cat $O/summary.txt
python - <<'PY'
import json
for f in ("bench_one_stream","bench_two_streams","bench_two_streams_graph"):
    try:
        d=json.loads([l for l in open("gpurun_out/r3r/%s.json"%f) if l.startswith("{")][0])
        print(f, round(d["value"]/1e6,2), round(d["ms_per_step"],3), "elim", (d.get("bg_elimination") or {}).get("ms_per_step"), "plain", (d.get("plain_f16_mode") or {}).get("ms_per_step"))
    except Exception as e: print(f, "ERR", e)
PY
