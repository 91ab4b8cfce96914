cd /root/repo
mkdir -p gpurun_out/r4e
T="timeout -k 10"
A="--train_steps 0 --variance 0.6 --v_jit 0.05"
$T 120 python scripts/diag/grad_breakdown.py $A > gpurun_out/r4e/g_both.log 2>&1
NEUCONW_NERF_RAY_BIAS=0 $T 120 python scripts/diag/grad_breakdown.py $A > gpurun_out/r4e/g_color_only.log 2>&1
NEUCONW_COLOR_RAY_BIAS=0 NEUCONW_NERF_RAY_BIAS=1 $T 120 python scripts/diag/grad_breakdown.py $A > gpurun_out/r4e/g_nerf_only.log 2>&1
NEUCONW_COLOR_RAY_BIAS=0 $T 120 python scripts/diag/grad_breakdown.py $A > gpurun_out/r4e/g_none.log 2>&1
NEUCONW_BG_DENSE=1 $T 120 python scripts/diag/grad_breakdown.py $A > gpurun_out/r4e/g_both_dense.log 2>&1
NEUCONW_BG_STREAM=0 $T 120 python scripts/diag/grad_breakdown.py $A > gpurun_out/r4e/g_both_1stream.log 2>&1
grep -h "^env\|embedding" gpurun_out/r4e/g_*.log
