"""CPU: `python bench.py --gpus N` (N > 1) launches its own N ranks (one process per GPU, train.py:53-55) and the
ranks reach init_process_group with that world size.  gloo stands in for RCCL here (no GPU in this container)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra):
    env = dict(os.environ, NCW_DIST_BACKEND="gloo", **env_extra)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env,
                       timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks():
    out = _run(["--gpus", "2", "--dist-check"], {})
    assert out["world"] == 2 and out["backend"] == "gloo"
    assert sorted(r["rank"] for r in out["ranks"]) == [0, 1]
    assert len({r["pid"] for r in out["ranks"]}) == 2  # two processes, not two threads


def test_bench_under_torchrun_does_not_relaunch():
    """The driver's form: torch.distributed.run starts the ranks; bench.py must use them as they are."""
    env = dict(os.environ, NCW_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29611", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-check"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["world"] == 2


def test_bench_rejects_world_mismatch():
    env = dict(os.environ, NCW_DIST_BACKEND="gloo", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-check"],
                       capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stdout + r.stderr)
