cd /root/repo
mkdir -p gpurun_out/r4g
T="timeout -k 10"
$T 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_train_driver.py tests/test_gpu_f16.py -q --timeout 400 -s > gpurun_out/r4g/t.log 2>&1; echo "t rc $?" >> gpurun_out/r4g/status
$T 100 python __graft_entry__.py --smoke > gpurun_out/r4g/smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/r4g/status
cat gpurun_out/r4g/status; grep -E "passed|failed" gpurun_out/r4g/t.log | tail -3; grep -E "^FAILED" gpurun_out/r4g/t.log
