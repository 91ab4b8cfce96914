cd /root/repo
mkdir -p gpurun_out/r4d
T="timeout -k 10"
$T 900 python -m pytest tests -m gpu -q --timeout 500 -s > gpurun_out/r4d/full.log 2>&1; echo "full rc $?" >> gpurun_out/r4d/status
$T 400 python bench.py > gpurun_out/r4d/bench.json 2> gpurun_out/r4d/bench.err; echo "bench rc $?" >> gpurun_out/r4d/status
cat gpurun_out/r4d/status; grep -E "passed|failed" gpurun_out/r4d/full.log | tail -3
