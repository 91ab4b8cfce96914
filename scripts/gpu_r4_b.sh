cd /root/repo
mkdir -p gpurun_out/r4b
T="timeout -k 10"
$T 300 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_color_nerf.py -x -q --timeout 200 -s > gpurun_out/r4b/t1.log 2>&1; echo "t1 rc $?" >> gpurun_out/r4b/status
$T 240 python -m pytest tests/test_gpu_rccl_world1.py -x -q --timeout 200 -s > gpurun_out/r4b/t2.log 2>&1; echo "t2 rc $?" >> gpurun_out/r4b/status
$T 200 python scripts/diag/pp_epilogue.py > gpurun_out/r4b/pp_epilogue.log 2>&1; echo "pp rc $?" >> gpurun_out/r4b/status
$T 400 python bench.py > gpurun_out/r4b/bench.json 2> gpurun_out/r4b/bench.err; echo "bench rc $?" >> gpurun_out/r4b/status
$T 500 python -m pytest tests/test_gpu_ddp.py -x -q --timeout 400 -s > gpurun_out/r4b/t3.log 2>&1; echo "t3 rc $?" >> gpurun_out/r4b/status
$T 300 python -m pytest tests/test_gpu_fullsize.py -x -q --timeout 200 -s -k "after_training or trained_operating" > gpurun_out/r4b/t4.log 2>&1; echo "t4 rc $?" >> gpurun_out/r4b/status
$T 300 python -m pytest tests/test_gpu_train_driver.py -x -q --timeout 250 -s > gpurun_out/r4b/t5.log 2>&1; echo "t5 rc $?" >> gpurun_out/r4b/status
$T 120 python - > gpurun_out/r4b/dump.log 2>&1 <<'P'
import torch, sys
sys.path.insert(0, "/root/repo")
from tests._parity import trained_weights
sd = trained_weights(256, 64, 64, 5, 0.0, 40)
torch.save({k: v.half() if False else v for k, v in sd.items()}, "/root/repo/gpurun_out/r4b/trained_w256.pt")
print("saved", sum(v.numel() for v in sd.values()))
P
echo "dump rc $?" >> gpurun_out/r4b/status
cat gpurun_out/r4b/status
