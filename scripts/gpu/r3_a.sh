#!/bin/bash
# round 3, GPU call A: new tests first (fail fast), the trained-operating-point table, bench with the parity object, graph replay
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3a
O=gpurun_out/r3a
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 900 python -m pytest tests/test_gpu_f16.py tests/test_gpu_trainer.py tests/test_gpu_ddp.py -x -q -s ) > $O/new_tests.log 2>&1
echo "new_tests rc=$?" >> $O/summary.txt
( time timeout 600 python scripts/diag/trained_point_parity.py --json $O/trained_point.json ) > $O/trained_point.log 2>&1
echo "trained_point rc=$?" >> $O/summary.txt
( time timeout 600 python bench.py ) > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" >> $O/summary.txt
( time timeout 300 python bench.py --graph --no-pmc --no-cpu-baseline --no-parity-mode ) > $O/bench_graph.json 2> $O/bench_graph.err
echo "bench_graph rc=$?" >> $O/summary.txt
( time timeout 300 python bench.py --graph --bg-eliminate --no-pmc --no-cpu-baseline --no-parity-mode ) > $O/bench_graph_elim.json 2> $O/bench_graph_elim.err
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/gpu_tests.log 2>&1
echo "gpu_tests rc=$?" >> $O/summary.txt
tail -3 $O/new_tests.log; tail -30 $O/trained_point.log; cat $O/summary.txt; tail -3 $O/gpu_tests.log
