cd /root/repo
mkdir -p gpurun_out/r4f
T="timeout -k 10"
$T 900 python -m pytest tests -m gpu -q --timeout 500 -s > gpurun_out/r4f/full.log 2>&1; echo "full rc $?" >> gpurun_out/r4f/status
$T 200 python __graft_entry__.py --smoke > gpurun_out/r4f/smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/r4f/status
$T 900 bash scripts/collect_profiles.sh v1 > gpurun_out/r4f/collect.log 2>&1; echo "collect rc $?" >> gpurun_out/r4f/status
cat gpurun_out/r4f/status; grep -E "passed|failed" gpurun_out/r4f/full.log | tail -3; grep -E "^FAILED" gpurun_out/r4f/full.log
