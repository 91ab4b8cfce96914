#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r3q
timeout 600 python scripts/diag/sampler_split.py > gpurun_out/r3q/sampler_split.log 2>&1
grep variance gpurun_out/r3q/sampler_split.log
