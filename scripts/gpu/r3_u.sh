#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3u; mkdir -p $O
for e in ${EXPS:-0 8}; do NCW_P16_EXP=$e timeout 300 python scripts/time_infer16.py 2>/dev/null | grep "bf16" | sed "s/^/EXP=$e /"; done | tee $O/exp.log
