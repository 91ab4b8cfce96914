#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
O=gpurun_out/r3l; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 900 python -m pytest tests/test_gpu_mesh.py tests/test_gpu_sdf_train.py tests/test_gpu_sdf.py -q -x ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/summary.txt
( bash scripts/collect_profiles.sh v1 ) > $O/collect.log 2>&1
echo "collect rc=$?" >> $O/summary.txt
cd $R
( time timeout 600 python bench.py --config voxel --no-pmc --no-parity-mode --no-cpu-baseline ) > gpurun_out/r03/bench_voxel_v1.json 2>> $O/err.log
( time timeout 600 python bench.py --config grid512 --prec f16 ) > gpurun_out/r03/bench_grid512_v1.json 2>> $O/err.log
( time timeout 600 python bench.py --config grid512 --grid-width 256 --prec f16 ) > gpurun_out/r03/bench_grid256_v1.json 2>> $O/err.log
( time timeout 600 python bench.py --bg-eliminate --no-parity-mode --no-cpu-baseline ) > gpurun_out/r03/bench_elim_v1.json 2>> $O/err.log
tail -3 $O/tests.log; cat $O/summary.txt; ls gpurun_out/r03
