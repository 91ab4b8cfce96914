#!/bin/bash
# Round-5 GPU session 1: the whole -m gpu suite on the new build, smoke, the default bench line, the new secondary rows
# (render / voxel / grid512 with parity), the shipped shape three times with and without the elimination (VERDICT weak 5), the
# atomics probe and the RCCL world-1 call sequence under NCCL_DEBUG=INFO.  Every stage under `timeout`, own log.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05a; mkdir -p $OUT; rm -f $OUT/status
T="timeout -k 10"
$T 900 python -m pytest tests -m gpu -q --timeout 500 -x > $OUT/full.log 2>&1; echo "pytest -m gpu rc $?" >> $OUT/status
$T 900 python -m pytest tests/test_gpu_render_only.py tests/test_gpu_voxel.py tests/test_gpu_grid.py tests/test_gpu_fullsize.py tests/test_gpu_trainer.py -m gpu -q --timeout 500 -s > $OUT/new_tests.log 2>&1; echo "new tests rc $?" >> $OUT/status
$T 120 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc $?" >> $OUT/status
$T 400 python bench.py --save-trained-state $OUT/trained_state_bench.pt > $OUT/bench_v1.json 2> $OUT/bench_v1.err; echo "bench rc $?" >> $OUT/status
$T 300 python bench.py --config render > $OUT/bench_render.json 2> $OUT/bench_render.err; echo "render rc $?" >> $OUT/status
$T 300 python bench.py --config render --bg-eliminate --no-pmc > $OUT/bench_render_elim.json 2> $OUT/bench_render_elim.err; echo "render elim rc $?" >> $OUT/status
$T 300 python bench.py --config voxel --no-pmc > $OUT/bench_voxel.json 2> $OUT/bench_voxel.err; echo "voxel rc $?" >> $OUT/status
$T 300 python bench.py --config grid512 --prec f16 > $OUT/bench_grid512_f16.json 2> $OUT/bench_grid512.err; echo "grid512 rc $?" >> $OUT/status
for i in 1 2 3; do
  $T 200 python bench.py --config shipped --no-pmc --no-cpu-baseline --no-parity-mode > $OUT/shipped_dense_$i.json 2>/dev/null; echo "shipped dense $i rc $?" >> $OUT/status
  $T 200 python bench.py --config shipped --no-pmc --no-cpu-baseline --no-parity-mode --bg-eliminate > $OUT/shipped_elim_$i.json 2>/dev/null; echo "shipped elim $i rc $?" >> $OUT/status
done
$T 120 scripts/probes/atomic_probe > $OUT/atomic_probe.log 2>&1; echo "atomic probe rc $?" >> $OUT/status
PORT=$((20000 + RANDOM % 20000))
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT HSA_ENABLE_IPC_MODE_LEGACY=0 NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,COLL \
  $T 200 python tests/_rccl_world1_worker.py > $OUT/rccl_world1_nccl_debug.log 2>&1; echo "rccl world1 rc $?" >> $OUT/status
cat $OUT/status; grep -E "passed|failed" $OUT/full.log | tail -2; grep -E "^FAILED|^ERROR" $OUT/full.log $OUT/new_tests.log | head; tail -3 $OUT/smoke.log
python scripts/show_bench.py < $OUT/bench_v1.json 2>/dev/null | head -40
