"""Data-parallel gradient exchange: ONE RCCL all-reduce of a flat fp32 gradient buffer per step
(the contract of the reference's PL `accelerator='ddp'`, train.py:53-55; 8.4 MB at W=256).
Rays shard across ranks with no data-path collective; parameters are replicated."""
import torch
import torch.distributed as dist


def allreduce_grads(params, world_size=None, group=None, flat_buffers=()):
    """Average .grad of `params` across ranks.  `flat_buffers`: persistent flat gradient buffers whose
    views ARE the .grad of (most of) the parameters (NeuconWRenderer.flat_grad_buffer()): they are
    all-reduced IN PLACE with one collective each -- no flatten / unflatten copies.  Remaining
    parameters (embedding, variance) travel in one small concatenated buffer.  Parameters without a
    gradient (the reference's dead layers) are skipped consistently on every rank."""
    if not dist.is_available() or not dist.is_initialized():
        return
    world_size = dist.get_world_size(group) if world_size is None else world_size
    if world_size == 1:
        return
    handles = []
    covered = []
    for fb in flat_buffers:
        if fb is None:
            continue
        handles.append((dist.all_reduce(fb, op=dist.ReduceOp.SUM, group=group, async_op=True), fb))
        covered.append((fb.data_ptr(), fb.data_ptr() + fb.numel() * fb.element_size()))

    def in_flat(g):
        a = g.data_ptr()
        return any(lo <= a < hi for lo, hi in covered)

    grads = [p.grad for p in params if p.grad is not None and not in_flat(p.grad)]
    for h, fb in handles:
        h.wait()
        fb.div_(world_size)
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(world_size)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


def broadcast_params(modules, src=0, group=None):
    """Make every rank start from rank `src`'s parameters (DDP's constructor-time broadcast)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for m in modules:
        for p in m.parameters():
            dist.broadcast(p.data, src=src, group=group)
