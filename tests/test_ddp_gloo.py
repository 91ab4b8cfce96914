"""Multi-process (world_size 2, gloo, CPU) check of the gradient exchange used by bench.py --gpus N and scripts/train.py
(trainer.FlatParams): one broadcast, one flat all-reduce per step, mean over ranks, replicas bit-identical."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker_flat(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from neuralrecon_w_amd.trainer import FlatParams

    torch.manual_seed(rank)  # replicas start different: FlatParams.broadcast makes them rank 0's
    lin, emb = torch.nn.Linear(5, 3), torch.nn.Embedding(6, 2)
    fp = FlatParams([emb, lin])
    fp.broadcast()
    opt = torch.optim.Adam([fp.flat], lr=1e-2, eps=1e-7)
    g = torch.Generator().manual_seed(100 + rank)  # each rank owns its shard of the batch
    for _ in range(2):
        x, idx = torch.randn(4, 5, generator=g), torch.randint(0, 6, (3,), generator=g)
        fp.zero_grad()
        (lin(x).sum() + emb(idx).pow(2).sum()).backward()
        local = fp.flat_grad.clone()
        fp.allreduce()
        torch.nn.utils.clip_grad_norm_([fp.flat], 0.99)
        opt.step()
    q.put((rank, fp.flat.detach().tolist(), local.tolist(), fp.flat_grad.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_train_step_world2():
    """The trainer's exchange: ONE all-reduce of the flat gradient, replicas stay bit-identical."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_flat, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, w0, l0, g0), (_, w1, l1, g1) = res
    T = torch.tensor
    assert torch.equal(T(w0), T(w1))
    assert torch.equal(T(g0), T(g1))
    assert not torch.equal(T(l0), T(l1))  # the shards really differed
