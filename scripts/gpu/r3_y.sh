#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3y; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_mesh_sparse.py tests/test_gpu_mesh.py -q -x ) > $O/tests.log 2>&1
echo "rc=$?"; tail -30 $O/tests.log
