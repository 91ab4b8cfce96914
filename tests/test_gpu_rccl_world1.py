"""GPU: the RCCL backend itself on the real device.  The lease has ONE GPU, so every multi-rank test runs over gloo; this
one initialises backend "nccl" (RCCL) with world size 1 and pushes the three collective call sites of the path through it
(tests/_rccl_world1_worker.py).  What it cannot show is wire time or a second rank -- only that the RCCL code path
(device_id init, ReduceOp.AVG, all_gather_into_tensor, stream ordering against our launches) runs and returns the right data."""
import json
import os
import socket
import subprocess
import sys

import pytest

from tests._util import ROOT

pytestmark = pytest.mark.gpu


def test_collective_call_sites_run_over_rccl_with_one_rank():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_world1_worker.py")], capture_output=True, text=True,
                       env=env, timeout=180, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    print(out)
    assert out["backend"] == "nccl" and out["world"] == 1
    assert out["allreduce_avg_of_one_rank_is_identity"] and out["allreduce_numel"] > 100000
    assert out["train_step_loss_finite"]
    assert out["sdf_grid_equal"]
    assert out["surface_selection_equal"] and out["surface_selection_points"] > 0
