#!/bin/bash
# One gpurun call that verifies a build on the MI355X box: the whole `-m gpu` suite, smoke(), the round's profile set
# (rocprofv3 kernel stats + PMC passes + the default bench line) and the secondary bench rows.  Every stage runs under
# `timeout` and writes its own log, so a hang costs one stage, not the call (NOTEBOOK R4.6).
#     gpurun --timeout 2400 -- 'bash scripts/gpu_verify.sh [tag]'      -> gpurun_out/verify/, gpurun_out/<round>/
set -u
cd "$(dirname "$0")/.."
TAG=${1:-v1}
OUT=gpurun_out/verify; mkdir -p $OUT; rm -f $OUT/status
T="timeout -k 10"
$T 1200 python -m pytest tests -m gpu -q -s --timeout 500 > $OUT/full.log 2>&1; echo "pytest -m gpu rc $?" >> $OUT/status
$T 100 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc $?" >> $OUT/status
NCW_PROFILE_ROUND=${NCW_PROFILE_ROUND:-r06} $T 900 bash scripts/collect_profiles.sh "$TAG" > $OUT/collect.log 2>&1; echo "collect_profiles rc $?" >> $OUT/status
$T 300 python bench.py --config shipped --no-pmc > $OUT/bench_shipped_2048rays_$TAG.json 2>/dev/null; echo "shipped rc $?" >> $OUT/status
$T 300 python bench.py --config voxel --no-pmc > $OUT/bench_voxel_$TAG.json 2>/dev/null; echo "voxel rc $?" >> $OUT/status
$T 300 python bench.py --bg-eliminate --no-cpu-baseline --no-parity-mode > $OUT/bench_elim_$TAG.json 2>/dev/null; echo "elim rc $?" >> $OUT/status
$T 300 python bench.py --config shipped --bg-eliminate --no-pmc --no-cpu-baseline --no-parity-mode > $OUT/bench_shipped_2048rays_elim_$TAG.json 2>/dev/null; echo "shipped elim rc $?" >> $OUT/status
$T 300 python bench.py --config render > $OUT/bench_render_$TAG.json 2>/dev/null; echo "render rc $?" >> $OUT/status
$T 300 python bench.py --config grid512 --prec f16 > $OUT/bench_grid512_f16_$TAG.json 2>/dev/null; echo "grid512 f16 (split value path) rc $?" >> $OUT/status
$T 300 python bench.py --config grid512 --prec bf16 --no-cpu-baseline > $OUT/bench_grid512_bf16_$TAG.json 2>/dev/null; echo "grid512 bf16 rc $?" >> $OUT/status
$T 400 python bench.py --rays 8192 --no-pmc > $OUT/bench_rays8192_1gpu_$TAG.json 2>/dev/null; echo "rays 8192 rc $?" >> $OUT/status
for N in 1500 3000; do
  $T 600 python scripts/diag/train_equivalence.py --steps $N --precs f32,f32b,f16,f16_noextras --out $OUT/train_equivalence_$N.json > $OUT/train_equivalence_$N.log 2>&1; echo "train_equivalence $N rc $?" >> $OUT/status
done
cat $OUT/status; grep -E "passed|failed" $OUT/full.log | tail -2; grep -E "^FAILED" $OUT/full.log; tail -3 $OUT/smoke.log
