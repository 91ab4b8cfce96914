"""A PL-free training loop on synthetic rays: the reference's recipe (train.py:16-64,
neuconw_system.py:142-184, 266-312, 376-400) without PyTorch-Lightning -- LR rule, Adam eps 1e-7, clip 0.99,
cos-anneal schedule, periodic octree refresh from the current SDF, checkpoints in the reference's state_dict layout.

    python scripts/train_synthetic.py --steps 200 --rays 1024 --ckpt /tmp/neuconw_synth.ckpt
(one process per GPU under `python -m torch.distributed.run --nproc-per-node N ...` shards the rays data-parallel)
"""
import argparse
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (model / batch builders of the benchmark: BASELINE configs[1])
import neuralrecon_w_amd as nw  # noqa: E402
from neuralrecon_w_amd import trainer, voxel  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--prec", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--update-freq", type=int, default=0, help="octree refresh period (NEUCONW.UPDATE_FREQ), 0 = off")
    ap.add_argument("--train-level", type=int, default=7)
    ap.add_argument("--ckpt", default="")
    ap.add_argument("--resume", default="")
    args = ap.parse_args()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    prec = {"bf16": nw.PREC_BF16, "f16": nw.PREC_F16, "f32": nw.PREC_F32}[args.prec]
    emb, neuconw, nerf, rdr = bench.build_models(dev, prec)
    if args.update_freq:  # a coarse occupancy shell around the initial surface (stands in for the SfM octree)
        G = 32
        c = (torch.stack(torch.meshgrid(*[torch.arange(G, device=dev)] * 3, indexing="ij"), -1).float() + 0.5) * (2.0 / G) - 1
        rdr.octree_data = voxel.occupancy_from_dense((c.norm(dim=-1) - 0.5).abs() < 0.2, torch.zeros(3, device=dev), 1.0)
    step_fn = nw.TrainStep(rdr, [emb, neuconw, nerf], bench.loss_fn, lr=1e-4 * world * args.rays / 4096.0, eps=1e-7,
                           clip=0.99, world_size=world)
    start = 0
    if args.resume:
        ck = trainer.load_checkpoint(args.resume, emb, neuconw, nerf, flat_params=step_fn.fp)
        start = int(ck.get("global_step", 0))
        if "optimizer_states" in ck:
            step_fn.opt.load_state_dict(ck["optimizer_states"][0], trainer.reference_param_order(emb, neuconw, nerf))
    bg = torch.zeros(1, 3, device=dev)
    t0 = time.perf_counter()
    for it in range(start, start + args.steps):
        rays, ts, label, rgbs = bench.synth_batch(args.rays, 1000 + rank + world * it, dev)  # a fresh batch per step
        loss, out = step_fn(rays, ts, label, rgbs, background_rgb=bg, cos_anneal_ratio=min(1.0, it / 50000.0))
        if args.update_freq and (it + 1) % args.update_freq == 0:
            data = voxel.octree_update(rdr, args.train_level, threshold=0.05)  # neuconw_system.py:266-312
            if rank == 0:
                print("step %d: octree refreshed, %d fine voxels" % (it + 1, int(voxel.dense_from_occupancy(data).sum())))
        if rank == 0 and (it % 20 == 0 or it == start + args.steps - 1):
            gn = float(getattr(step_fn, "last_grad_norm", float("nan")))
            print("step %5d  loss %.5f  s_val %.4f  |grad| %.4g%s" % (it, float(loss), float(out["s_val"]), gn,
                  "" if gn == gn and abs(gn) != float("inf") else "  <- non-finite: update skipped (fp16 overflow? try --prec bf16)"))
    torch.cuda.synchronize()
    if rank == 0:
        dt = time.perf_counter() - t0
        print("%d steps, %.2f ms/step incl. batch synthesis and logging" % (args.steps, 1e3 * dt / args.steps))
        if args.ckpt:
            trainer.save_checkpoint(args.ckpt, emb, neuconw, nerf, optimizer=step_fn.opt, global_step=start + args.steps)
            print("wrote", args.ckpt)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
