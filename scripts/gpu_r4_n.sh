cd /root/repo
mkdir -p gpurun_out/r4n
T="timeout -k 10"
$T 900 python -m pytest tests -m gpu -q --timeout 500 > gpurun_out/r4n/full.log 2>&1; echo "full rc $?" >> gpurun_out/r4n/status
$T 100 python __graft_entry__.py --smoke > gpurun_out/r4n/smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/r4n/status
NCW_PROFILE_ROUND=r04 $T 900 bash scripts/collect_profiles.sh v2 > gpurun_out/r4n/collect.log 2>&1; echo "collect rc $?" >> gpurun_out/r4n/status
$T 300 python bench.py --config shipped --no-pmc > gpurun_out/r4n/bench_shipped_2048rays_v2.json 2>/dev/null; echo "shipped rc $?" >> gpurun_out/r4n/status
$T 300 python bench.py --config voxel --no-pmc > gpurun_out/r4n/bench_voxel_v2.json 2>/dev/null; echo "voxel rc $?" >> gpurun_out/r4n/status
$T 300 python bench.py --bg-eliminate --no-cpu-baseline --no-parity-mode > gpurun_out/r4n/bench_elim_v2.json 2>/dev/null; echo "elim rc $?" >> gpurun_out/r4n/status
cat gpurun_out/r4n/status; grep -E "passed|failed" gpurun_out/r4n/full.log | tail -2; grep -E "^FAILED" gpurun_out/r4n/full.log; tail -3 gpurun_out/r4n/smoke.log
