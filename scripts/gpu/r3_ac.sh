#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3ac; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_grid.py tests/test_gpu_mesh.py tests/test_gpu_mesh_sparse.py tests/test_gpu_octree_refresh.py tests/test_gpu_sdf.py tests/test_gpu_voxel.py tests/test_gpu_train_driver.py -q -x ) > $O/tests.log 2>&1
echo "rc=$?"; tail -5 $O/tests.log
timeout 600 python bench.py --config grid512 --prec f32 --steps 1 --warmup 1 2>/dev/null | tail -c 600 > $O/grid512_f32.json; python -c "
import json;d=json.loads(open('$O/grid512_f32.json').read()[open('$O/grid512_f32.json').read().index('{'):]) if False else None" 2>/dev/null
timeout 600 python bench.py --config grid512 --prec f32 --steps 1 --warmup 1 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('grid512 f32:', d['value']/1e6, 'Mpts/s', d['ms_per_step'], 'ms')"
