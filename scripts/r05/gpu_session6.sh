#!/bin/bash
# Round-5 GPU session 6: full suite on the tree with the large-ray per-ray kernels (> 512 samples per ray) and the coherent-error normals test;
# the default bench line again (the per-ray kernels moved into a namespace: same code).
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05g; mkdir -p $OUT; rm -f $OUT/status
T="timeout -k 10"
$T 900 python -m pytest tests -m gpu -q --timeout 500 > $OUT/full.log 2>&1; echo "pytest -m gpu rc $?" >> $OUT/status
$T 300 python -m pytest tests/test_gpu_sdf.py tests/test_gpu_edges.py -m gpu -q -s -k "adjoint or 512" > $OUT/new_tests.log 2>&1; echo "new tests rc $?" >> $OUT/status
$T 100 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc $?" >> $OUT/status
$T 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench.err; echo "bench rc $?" >> $OUT/status
cat $OUT/status; grep -E "passed|failed" $OUT/full.log | tail -2; grep -E "^FAILED|^ERROR" $OUT/full.log | head; grep -E "normals vs fp64|samples per ray" $OUT/new_tests.log
