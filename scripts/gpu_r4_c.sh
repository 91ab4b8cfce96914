cd /root/repo
mkdir -p gpurun_out/r4c
T="timeout -k 10"
$T 200 python scripts/diag/grad_breakdown.py > gpurun_out/r4c/grad_on.log 2>&1; echo "g1 rc $?" >> gpurun_out/r4c/status
NEUCONW_COLOR_RAY_BIAS=0 $T 200 python scripts/diag/grad_breakdown.py > gpurun_out/r4c/grad_off.log 2>&1; echo "g2 rc $?" >> gpurun_out/r4c/status
$T 300 python -m pytest tests/test_gpu_color_nerf.py -x -q --timeout 200 -s > gpurun_out/r4c/t1.log 2>&1; echo "t1 rc $?" >> gpurun_out/r4c/status
$T 500 python -m pytest tests/test_gpu_ddp.py -x -q --timeout 400 -s -k "complete" > gpurun_out/r4c/t3.log 2>&1; echo "t3 rc $?" >> gpurun_out/r4c/status
$T 400 python -m pytest tests/test_gpu_train_driver.py -x -q --timeout 300 -s > gpurun_out/r4c/t5.log 2>&1; echo "t5 rc $?" >> gpurun_out/r4c/status
$T 300 python scripts/diag/pp_epilogue_pmc.py gpurun_out/r4c/pp_epilogue_pmc.json > gpurun_out/r4c/pp_pmc.log 2>&1; echo "pmc rc $?" >> gpurun_out/r4c/status
cat gpurun_out/r4c/status
