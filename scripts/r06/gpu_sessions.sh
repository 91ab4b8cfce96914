#!/bin/bash
# Round-6 GPU sessions, one function per gpurun call:   gpurun --timeout T -- 'bash scripts/r06/gpu_sessions.sh <name>'
# Every session writes under gpurun_out/r06_<name>/ ; what is judged is copied into profiles/r06/ afterwards.
set -u
cd "$(dirname "$0")/../.."
NAME=${1:?session name}
OUT=gpurun_out/r06_$NAME; mkdir -p $OUT
export TMPDIR=/tmp
B="python bench.py --no-pmc --no-parity-mode --no-cpu-baseline"

s1() {  # trained-weights parity point of the SHIPPED shape for the CPU emulation; config 4's whole batch on one GPU; training equivalence
  for S in 1000 2000 3000; do
    timeout -k 10 300 $B --config shipped --seed $S --save-trained-state $OUT/trained_shipped_seed$S.pt > $OUT/bench_shipped_seed$S.json 2>$OUT/bench_shipped_seed$S.err; echo "shipped seed $S rc $?"
  done
  timeout -k 10 400 python bench.py --rays 8192 --no-pmc > $OUT/bench_rays8192.json 2>$OUT/bench_rays8192.err; echo "rays 8192 rc $?"
  for N in 1500 3000; do
    timeout -k 10 600 python scripts/diag/train_equivalence.py --steps $N --precs f32,f32b,f16,f16_noextras --out $OUT/train_equivalence_$N.json > $OUT/train_equivalence_$N.log 2>&1; echo "train_equivalence $N rc $?"
  done
}

s2() {  # (s1 skipped the parity legs with --no-cpu-baseline: the trained state is written by the parity leg)
  for S in 1000 2000 3000; do
    timeout -k 10 400 python bench.py --no-pmc --no-parity-mode --config shipped --seed $S --save-trained-state $OUT/trained_shipped_seed$S.pt > $OUT/bench_shipped_seed$S.json 2>$OUT/bench_shipped_seed$S.err; echo "shipped seed $S rc $?"
  done
}

s3() {  # s2 again on a consistent build + the new voxel / kaolin-boundary tests
  s2
  timeout -k 10 900 python -m pytest tests/test_gpu_voxel.py tests/test_gpu_octree_refresh.py tests/test_gpu_mesh.py tests/test_gpu_trainer.py tests/test_compat_kaolin.py tests/test_octree_from_sfm.py -x -q -s -m "gpu or not gpu" > $OUT/voxel_tests.log 2>&1; echo "voxel tests rc $?"
  tail -5 $OUT/voxel_tests.log
}

"$NAME"
ls -la $OUT
