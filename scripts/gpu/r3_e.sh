#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3e; mkdir -p $O
( NCW_SPLIT_V=1 timeout 300 python scripts/diag/split_check.py ) > $O/split_check_v1.log 2>&1
( timeout 300 python scripts/diag/split_check.py ) > $O/split_check_v2.log 2>&1
( timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_sdf.py tests/test_gpu_sdf_train.py -q -x ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/summary.txt
( timeout 600 python bench.py --no-pmc --no-parity-mode --no-cpu-baseline ) > $O/bench.json 2> $O/bench.err
grep "f16 split" $O/split_check_v1.log $O/split_check_v2.log; tail -3 $O/tests.log; cat $O/summary.txt
python -c "
import json
d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][0])
print(d['value']/1e6, d['ms_per_step'], d['roofline']['per_step_kernel_ms'])"
