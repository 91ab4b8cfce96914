#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for rep in 1 2; do
timeout 300 python scripts/time_sdf_infer.py 2>/dev/null | grep "W=256 bf16" | sed "s/^/base /"
NEUCONW_HIP_LIB=$PWD/neuralrecon-w_amd/libneuconw_hip_ppst.so timeout 300 python scripts/time_sdf_infer.py 2>/dev/null | grep "W=256 bf16" | sed "s/^/with stores /"
done
