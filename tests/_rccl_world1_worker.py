"""Worker of tests/test_gpu_rccl_world1.py: ONE rank, backend "nccl" (= RCCL on ROCm) on the real device.  A one-rank group
is legal and runs the same code path as N ranks up to the wire: init_process_group(device_id=...), communicator creation,
ReduceOp.AVG, all_gather_into_tensor, the stream hand-off between RCCL's stream and the launch stream.  Pushed through it:
trainer.FlatParams.allreduce (train.py:53-55), grid.sdf_grid (utils/visualization.py:27-35,81-83) and
voxel.surface_selection (neuconw_system.py:253-258), each compared with its no-collective result.  Prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)  # RANK / WORLD_SIZE / MASTER_* from the environment (world 1)
    import neuralrecon_w_amd as nw
    from neuralrecon_w_amd import grid, voxel
    from neuralrecon_w_amd.trainer import FlatParams
    from tests._build import build_system, loss_from_outputs
    from tests._util import synth_rays

    res = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    emb, neuconw, nerf, rdr = build_system(seed=3, prec=nw.PREC_F32)
    # ---- 1. the flat gradient all-reduce (ReduceOp.AVG, in place, async handle) ---------------------------------------
    fp = FlatParams([emb, neuconw, nerf], rdr)
    torch.manual_seed(0)
    fp.flat_grad.copy_(torch.randn_like(fp.flat_grad))
    before = fp.flat_grad.clone()
    h = fp.allreduce(async_op=True, force=True)
    assert h is not None
    h.wait()
    torch.cuda.synchronize()
    res["allreduce_avg_of_one_rank_is_identity"] = bool(torch.equal(fp.flat_grad, before))
    res["allreduce_numel"] = fp.flat_grad.numel()
    # ... and a whole TrainStep on top of an initialised RCCL group (world_size 1: no exchange, nothing may break)
    train = nw.TrainStep(rdr, [emb, neuconw, nerf], loss_from_outputs, lr=1e-3, eps=1e-7, clip=0.99, world_size=1)
    rays, ts, label, rgbs = [t.cuda() for t in synth_rays(32, 5, 64)]
    loss, _ = train(rays, ts, label, rgbs, background_rgb=torch.zeros(1, 3).cuda(), cos_anneal_ratio=0.2, perturb_overwrite=0)
    res["train_step_loss_finite"] = bool(torch.isfinite(loss))
    # ---- 2. config 5: the grid sweep assembled by all_gather_into_tensor ----------------------------------------------
    dim = 40
    a = grid.sdf_grid(neuconw.sdf_net, dim, force_collective=True)
    b = grid.sdf_grid_range(neuconw.sdf_net, dim, (-1.0,) * 3, (1.0,) * 3, 0, dim ** 3).view(dim, dim, dim)
    res["sdf_grid_equal"] = bool(torch.equal(a, b))
    # ---- 3. N1: the octree refresh's sharded SDF sweep + all_gather --------------------------------------------------
    occ = torch.zeros(16, 16, 16, dtype=torch.bool, device=dev)
    occ[4:12, 4:12, 4:12] = True
    rdr.octree_data = voxel.occupancy_from_dense(occ, torch.zeros(3), 1.0, voxel_size=2.0 / 16)
    p1, v1 = voxel.surface_selection(rdr, 6, 0.05, force_collective=True)
    p0, v0 = voxel.surface_selection(rdr, 6, 0.05)
    res["surface_selection_equal"] = bool(torch.equal(p0, p1)) and v0 == v1
    res["surface_selection_points"] = int(p1.shape[0])
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
