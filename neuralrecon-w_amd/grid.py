"""Config 5: SDF evaluation on a dense regular grid for marching cubes (tools/extract_mesh.py,
utils/visualization.py:37-85), sharded over the ranks of one node.

The reference builds the `[dim^3, 3]` coordinate array on the CPU, splits it per rank, copies each
chunk to the GPU, runs `renderer.sdf`, copies back, and all-gathers.  Here the coordinates are
generated on chip from the linear index (no coordinate tensor exists), each rank sweeps its
contiguous range with the fused SDF kernel, and one `all_gather` of fp32 SDF values assembles the
grid.  (Marching cubes itself is the 'next' row N2 of SURVEY 8f and not part of this path.)
"""
import torch
import torch.distributed as dist

from . import lib as L


def local_range(total, rank, world):
    """Contiguous equal split, padded like utils/visualization.py:27-35 (get_local_split)."""
    per = (total + world - 1) // world
    return rank * per, min(per, max(0, total - rank * per)), per


@torch.no_grad()
def sdf_grid_range(sdf_net, dim, bound_min, bound_max, start, count, origin=(0.0, 0.0, 0.0), radius=1.0, prec=None,
                   chunk=1 << 24, out=None):
    """sdf of grid points [start, start+count) of linspace(bound_min, bound_max, dim)^3 ('ij', x slowest)."""
    prec = sdf_net.value_prec() if prec is None else prec
    dev = next(sdf_net.parameters()).device
    if dev.type != "cuda":
        raise L.NeuconwHipError("sdf_grid needs the network on a GPU")
    plan = sdf_net.packed(prec)
    out = torch.empty(count, device=dev, dtype=torch.float32) if out is None else out
    pts = L.NcwPoints()
    pts.mode, pts.per_ray, pts.gdim, pts.gradius = 3, 1, int(dim), float(radius)
    for a in range(3):
        pts.gmin[a], pts.gmax[a], pts.gorigin[a] = float(bound_min[a]), float(bound_max[a]), float(origin[a])
    lib = L.get_lib()
    done = 0
    while done < count:
        n = min(chunk, count - done)
        pts.gstart = int(start + done)
        L.check(lib.ncw_sdf_infer_points(plan.net, prec, pts, n, out.data_ptr() + 4 * done, L.stream_ptr(dev)),
                "ncw_sdf_infer_points")
        done += n
    return out


@torch.no_grad()
def sdf_grid(sdf_net, dim, bound_min=(-1.0, -1.0, -1.0), bound_max=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0),
             radius=1.0, prec=None, group=None, force_collective=False):
    """Full [dim, dim, dim] SDF grid; with torch.distributed initialised every rank evaluates its
    contiguous 1/world slice and one all_gather assembles the result on all ranks (utils/visualization.py:27-35,81-83).
    `force_collective`: issue the all_gather in a ONE-rank group too (tests push the RCCL path through a single GPU)."""
    total = dim ** 3
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force_collective):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        start, count, per = local_range(total, rank, world)
        dev = next(sdf_net.parameters()).device
        local = torch.zeros(per, device=dev, dtype=torch.float32)
        if count > 0:
            sdf_grid_range(sdf_net, dim, bound_min, bound_max, start, count, origin, radius, prec, out=local)
        if dist.get_backend(group) == "nccl":  # RCCL: one flat all-gather straight into the result
            full = torch.empty(per * world, device=dev, dtype=torch.float32)
            dist.all_gather_into_tensor(full, local, group=group)
        else:  # gloo (CPU tests, ranks sharing one GPU): the list form
            parts = [torch.empty_like(local) for _ in range(world)]
            dist.all_gather(parts, local, group=group)
            full = torch.cat(parts)
        return full[:total].view(dim, dim, dim)
    return sdf_grid_range(sdf_net, dim, bound_min, bound_max, 0, total, origin, radius, prec).view(dim, dim, dim)
