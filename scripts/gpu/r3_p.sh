#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
O=gpurun_out/r3p; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( bash scripts/collect_profiles.sh v2 ) > $O/collect.log 2>&1
echo "collect rc=$?" >> $O/summary.txt
cd $R
( timeout 600 python bench.py --config voxel --no-pmc --no-parity-mode --no-cpu-baseline ) > gpurun_out/r03/bench_voxel_v2.json 2>> $O/err.log
( timeout 600 python bench.py --config grid512 --prec f16 ) > gpurun_out/r03/bench_grid512_v2.json 2>> $O/err.log
( timeout 600 python bench.py --config grid512 --grid-width 256 --prec f16 ) > gpurun_out/r03/bench_grid256_v2.json 2>> $O/err.log
( timeout 600 python bench.py --bg-eliminate --no-parity-mode --no-cpu-baseline ) > gpurun_out/r03/bench_elim_v2.json 2>> $O/err.log
( timeout 600 python bench.py --config shipped ) > gpurun_out/r03/bench_shipped_v2.json 2>> $O/err.log
( timeout 600 python bench.py --graph --no-pmc --no-cpu-baseline --no-parity-mode ) > gpurun_out/r03/bench_graph_v2.json 2>> $O/err.log
cat $O/summary.txt; ls gpurun_out/r03 | wc -l
