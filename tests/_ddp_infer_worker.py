"""Worker of tests/test_gpu_ddp.py::test_inference_collectives_with_two_ranks: rank r of a world-size-2 job whose ranks SHARE
GPU 0 (gloo carries the collectives).  The two inference-side exchange steps of the path at world > 1:

  * config 5, grid.sdf_grid (utils/visualization.py:27-35,81-83): each rank sweeps its contiguous padded slice of a dim^3
    lattice, one all_gather assembles it -- must equal the single-process sweep bit for bit, for a dim^3 that does NOT divide
    by the world size (the padded tail);
  * N1, voxel.surface_selection (neuconw_system.py:236-258): sharded SDF sweep over the up-sampled occupied voxels + all_gather
    -- the selected point set must equal the single-process selection.

Rank 0 prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    import neuralrecon_w_amd as nw
    from neuralrecon_w_amd import grid, voxel
    from tests._build import build_system

    emb, neuconw, nerf, rdr = build_system(seed=3, prec=nw.PREC_F32)  # same seed on both ranks: replicated weights
    res = {"world": world}
    dim = 37  # 37^3 = 50653 is odd: the last rank's slice has a padded tail
    full = grid.sdf_grid(neuconw.sdf_net, dim)
    single = grid.sdf_grid_range(neuconw.sdf_net, dim, (-1.0,) * 3, (1.0,) * 3, 0, dim ** 3).view(dim, dim, dim)
    start, count, per = grid.local_range(dim ** 3, rank, world)
    res["grid_equal"] = bool(torch.equal(full, single))
    res["grid_slices"] = [start, count, per]
    occ = torch.zeros(16, 16, 16, dtype=torch.bool, device="cuda")
    occ[3:12, 4:12, 4:13] = True  # 9 * 8 * 9 = 648 voxels x 64 sub-voxels: not a multiple of every chunk either
    rdr.octree_data = voxel.occupancy_from_dense(occ, torch.zeros(3), 1.0, voxel_size=2.0 / 16)
    pts, vs = voxel.surface_selection(rdr, 6, 0.03, chunk=7001)
    # the single-process result: the same call with the process group hidden from it
    n_local = voxel._sdf_sharded.__defaults__  # noqa: F841  (documentation: group=None -> default group)
    xyz_all, _ = voxel.surface_selection(rdr, 6, 1e9, chunk=7001)  # every candidate (threshold +inf), sharded
    sdf_single = rdr.sdf(((xyz_all - rdr.origin.float().cuda().reshape(3)) / rdr.radius).reshape(-1, 1, 3)).reshape(-1)
    want = xyz_all[sdf_single <= 0.03]
    res["selection_equal"] = bool(torch.equal(pts, want))
    res["selection_points"], res["candidates"] = int(pts.shape[0]), int(xyz_all.shape[0])
    flags = torch.tensor([int(res["grid_equal"]), int(res["selection_equal"])])
    dist.all_reduce(flags, op=dist.ReduceOp.MIN)  # every rank must agree
    res["all_ranks_ok"] = bool(flags.min() == 1)
    torch.cuda.synchronize()
    dist.barrier()
    if rank == 0:
        print(json.dumps(res))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
