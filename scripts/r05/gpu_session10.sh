#!/bin/bash
# Round-5 GPU session 10: where the step's wall time is not covered by kernels: kernel trace of the inner loop (union of the
# kernel intervals per step against the step's span), and the same step replayed as a HIP graph.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05n; mkdir -p $OUT; rm -f $OUT/status
T="timeout -k 10"
export TMPDIR=/tmp
$T 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o k -- python bench.py --inner --no-pmc --no-cpu-baseline --no-parity-mode > $OUT/prof_bench.json 2> $OUT/prof.err; echo "prof rc $?" >> $OUT/status
find $OUT/prof -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace.csv \;
rm -rf $OUT/prof
python scripts/diag/trace_gaps.py $OUT/kernel_trace.csv > $OUT/trace_gaps.log 2>&1; echo "gaps rc $?" >> $OUT/status
rm -f $OUT/kernel_trace.csv
$T 200 python bench.py --graph --no-pmc --no-parity-mode --no-cpu-baseline > $OUT/bench_graph.json 2>$OUT/graph.err; echo "graph rc $?" >> $OUT/status
$T 200 python bench.py --no-pmc --no-parity-mode --no-cpu-baseline > $OUT/bench_eager.json 2>/dev/null; echo "eager rc $?" >> $OUT/status
cat $OUT/status; cat $OUT/trace_gaps.log | tail -60
python - <<'P'
import json
for f in ('bench_graph','bench_eager'):
    try:
        d=json.loads(open('gpurun_out/r05n/%s.json'%f).read().strip().splitlines()[-1]); print(f, round(d['ms_per_step'],3))
    except Exception as e: print(f,'ERR',e)
P
tail -3 $OUT/graph.err
