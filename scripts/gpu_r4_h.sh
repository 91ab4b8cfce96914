cd /root/repo
mkdir -p gpurun_out/r4h
T="timeout -k 10"
$T 900 python -m pytest tests -m gpu -q --timeout 500 > gpurun_out/r4h/full.log 2>&1; echo "full rc $?" >> gpurun_out/r4h/status
$T 300 python bench.py --config shipped --no-pmc > gpurun_out/r4h/bench_shipped_2048rays.json 2> gpurun_out/r4h/e1; echo "shipped rc $?" >> gpurun_out/r4h/status
$T 300 python bench.py --config voxel --no-pmc > gpurun_out/r4h/bench_voxel.json 2> gpurun_out/r4h/e2; echo "voxel rc $?" >> gpurun_out/r4h/status
$T 200 python bench.py --config grid512 > gpurun_out/r4h/bench_grid512.json 2> gpurun_out/r4h/e3; echo "grid rc $?" >> gpurun_out/r4h/status
cat gpurun_out/r4h/status; grep -E "passed|failed" gpurun_out/r4h/full.log | tail -2; grep -E "^FAILED" gpurun_out/r4h/full.log
