#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3i; mkdir -p $O
for R in 2 3 4; do
  ( NCW_S2_RING=$R timeout 300 python scripts/diag/split_check.py ) 2>&1 | grep "f16 split" | sed "s/^/ring $R default-flags: /" >> $O/rings.log
  ( NCW_S2_RING=$R NEUCONW_HIP_LIB=$PWD/neuralrecon-w_amd/libneuconw_hip_flags.so timeout 300 python scripts/diag/split_check.py ) 2>&1 | grep "f16 split" | sed "s/^/ring $R mlp-flags: /" >> $O/rings.log
done
cat $O/rings.log
