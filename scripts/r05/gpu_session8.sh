#!/bin/bash
# Round-5 GPU session 8: the colour network's lin0 with [points | normals] as hi + lo pairs (third ring pass) on top of the background
# refinement: the suites it touches, the bench line over four batch seeds, the kernel table of the headline command.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05l; mkdir -p $OUT; rm -f $OUT/status
T="timeout -k 10"
$T 700 python -m pytest tests/test_gpu_bg_select.py tests/test_gpu_color_nerf.py tests/test_gpu_fullsize.py tests/test_gpu_render_only.py tests/test_gpu_voxel.py tests/test_gpu_grid.py tests/test_gpu_render.py -m gpu -q --timeout 600 -s > $OUT/tests.log 2>&1; echo "tests rc $?" >> $OUT/status
for S in 1000 2000 3000 4000; do
  $T 200 python bench.py --seed $S --no-pmc --no-parity-mode > $OUT/bench_seed$S.json 2>/dev/null; echo "seed $S rc $?" >> $OUT/status
done
$T 200 python bench.py --no-pmc --no-cpu-baseline --no-parity-mode --bg-eliminate > $OUT/bench_elim.json 2>/dev/null; echo "elim rc $?" >> $OUT/status
export TMPDIR=/tmp
$T 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o k -- python bench.py --inner --no-pmc --no-cpu-baseline --no-parity-mode > $OUT/prof_bench.json 2> $OUT/prof.err; echo "prof rc $?" >> $OUT/status
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/prof
cat $OUT/status; grep -E "passed|failed" $OUT/tests.log | tail -2; grep -E "^FAILED|^ERROR" $OUT/tests.log | head; grep -E "background NeRF at the" $OUT/tests.log
head -25 $OUT/kernel_stats.csv | cut -c1-200
