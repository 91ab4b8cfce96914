"""PL-free training driver on a real scene (SURVEY 8f N4 + N3): the reference's `train.py` / `NeuconWSystem` recipe
(train.py:16-64, neuconw_system.py:60-184, 266-400) without PyTorch-Lightning 1.4.8 -- one process per GPU.

    python scripts/train.py --cfg_path config/train_brandenburg_gate.yaml --root_dir data/heritage-recon/brandenburg_gate \
        --batch_size 2048 --num_epochs 20 --exp_name bg            (N GPUs: python -m torch.distributed.run --nproc-per-node N ...)

  * experiment yaml + scene config.yaml as the reference reads them (neuralrecon_w_amd.config);
  * this rank's share of the npz ray-cache chunks (datasets/data.py:83-119) uploaded to HBM once, batches assembled on
    the device, black-listed labels (RAY_MASK_LIST) removed once at load time -> fixed-size, sync-free steps;
  * Adam(eps 1e-7), lr = CANONICAL_LR * world * batch / CANONICAL_BS, grad-norm clip 0.99, ONE flat RCCL all-reduce;
  * cos_anneal_ratio = min(1, step / ANNEAL_END); every UPDATE_FREQ steps the fine octree is rebuilt from the current SDF
    (coarse octree from the COLMAP points: voxel.octree_from_sfm); checkpoints every SAVE_FREQ steps in the reference's
    PyTorch-Lightning layout (state_dict keys + torch.optim.Adam optimizer state), readable by its load_ckpt;
  * LR_SCHEDULER none / cosine / steplr stepped per epoch like PL steps the reference's scheduler (config.lr_at_epoch).
"""
import argparse
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neuralrecon_w_amd as nw  # noqa: E402
from neuralrecon_w_amd import config as C  # noqa: E402
from neuralrecon_w_amd import raycache, trainer, voxel  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg_path", required=True)
    ap.add_argument("--root_dir", default=None, help="overrides DATASET.ROOT_DIR")
    ap.add_argument("--batch_size", type=int, default=2048, help="rays per GPU per step (scripts/train.sh: 2048)")
    ap.add_argument("--num_epochs", type=int, default=20)
    ap.add_argument("--max_steps", type=int, default=0, help="stop after this many steps (0 = run the epochs out)")
    ap.add_argument("--exp_name", default="exp")
    ap.add_argument("--prec", default=None, choices=["bf16", "f16", "f32"],
                    help="training precision; default = the package default (NEUCONW_PREC, f16: neuconw.default_prec)")
    ap.add_argument("--ckpt_path", default="", help="resume from this checkpoint")
    ap.add_argument("--log_every", type=int, default=100)
    args = ap.parse_args()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("NCW_TRAIN_ONE_GPU_TEST"):  # plumbing test: the ranks share GPU 0 (collectives over gloo)
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("NCW_DIST_BACKEND", "nccl")  # nccl = RCCL over xGMI
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local)
    cfg = C.load_config(args.cfg_path, {"DATASET": {"ROOT_DIR": args.root_dir}} if args.root_dir else None)
    lr = C.scale_lr(cfg, world, args.batch_size)
    torch.manual_seed(cfg["TRAINER"]["SEED"])  # pl.seed_everything (train.py:18): identical initial weights on every rank
    prec = {None: None, "bf16": nw.PREC_BF16, "f16": nw.PREC_F16, "f32": nw.PREC_F32}[args.prec]
    emb, neuconw, nerf, rdr, scene = C.build_system(cfg, dev, prec)
    rdr.sync_free = True
    n, pt = cfg["NEUCONW"], cfg["DATASET"]["PHOTOTOURISM"]
    root = cfg["DATASET"]["ROOT_DIR"]
    names = raycache.local_splits(raycache.list_splits(root, pt["CACHE_DIR"]), world, rank)
    cache = raycache.RayCache(root, pt["CACHE_DIR"], names, dev, img_downscale=pt["IMG_DOWNSCALE"],
                              with_semantics=pt["WITH_SEMANTICS"], ray_mask_list=n["RAY_MASK_LIST"], prefilter=True)
    # Every rank runs the SAME number of steps per epoch: with the black-listed rays removed at load time the ranks'
    # caches differ in length, and a rank that leaves the loop early would strand the others in the gradient all-reduce
    # (the reference pads its chunks to equal length and filters inside the batch: datasets/data.py:83-119).
    steps_per_epoch = len(cache) // args.batch_size
    if world > 1:
        t = torch.tensor([steps_per_epoch], device=dev, dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        steps_per_epoch = int(t.item())
    if steps_per_epoch < 1:
        raise SystemExit("a rank holds fewer than --batch_size rays (%d)" % len(cache))
    if rank == 0:
        print("[rank 0] %d cache chunk(s), %d rays resident on %s; %d steps/epoch; lr %.3g"
              % (len(names), len(cache), dev, steps_per_epoch, lr))
    step_fn = nw.TrainStep(rdr, [emb, neuconw, nerf], C.neuconw_loss(cfg), lr=lr, eps=1e-7, clip=0.99, world_size=world)
    order = trainer.reference_param_order(emb, neuconw, nerf)
    step, resume = 0, None
    if args.ckpt_path:
        ck = trainer.load_checkpoint(args.ckpt_path, emb, neuconw, nerf, flat_params=step_fn.fp)
        step = int(ck.get("global_step", 0))
        if ck.get("optimizer_states"):
            step_fn.opt.load_state_dict(ck["optimizer_states"][0], order)
        resume = ck.get("ncw_resume")
    update_freq = int(n["UPDATE_FREQ"])
    train_level = C.surface_level(n["TRAIN_VOXEL_SIZE"], scene["eval_bbx"]) if update_freq > 0 else None
    save_dir = os.path.join(cfg["TRAINER"]["SAVE_DIR"], args.exp_name)
    bg = torch.zeros(1, 3, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(cfg["TRAINER"]["SEED"] + rank)
    t0, t_last, done = time.perf_counter(), time.perf_counter(), False
    # Resume: continue the interrupted epoch where it stopped -- same permutation (the generator goes back to its state at
    # the start of that epoch), the consumed batches skipped, the fp16 loss scale and its counters restored -- so a resumed
    # run executes the same steps as an uninterrupted one.  (Checkpoints without `ncw_resume`, e.g. the reference's own:
    # the epoch restarts from its first batch with a fresh generator, as before.)
    first_epoch, skip = step // steps_per_epoch, 0
    if resume is not None:
        first_epoch, skip, gstate = trainer.apply_resume_state(resume, step_fn.opt, rdr)
        if world > 1:  # per-rank generator states are not in rank 0's checkpoint: only rank 0 continues its permutation
            gstate = gstate if rank == 0 else None
        if gstate is not None:
            gen.set_state(gstate.to(gen.get_state().device))
        if skip >= steps_per_epoch:  # the checkpoint was written after the LAST batch of its epoch: consume that epoch's
            torch.randperm(len(cache), device=dev, generator=gen)  # permutation draw, the next epoch then draws its own
            first_epoch, skip = first_epoch + 1, 0
    if resume is not None and update_freq > 0 and step >= update_freq:
        # the fine octree is not part of a checkpoint: rebuild it from the restored SDF (identical to the uninterrupted run's
        # when the checkpoint was written right after a refresh, i.e. at a multiple of UPDATE_FREQ steps)
        if rdr.octree_data is None:  # normally built by the first render (NEAR_FAR_OVERRIDE) -- here before any
            rdr.octree_data = rdr.get_octree(dev)
        voxel.octree_update(rdr, train_level, n["SDF_THRESHOLD"])
    sched_kind = cfg["TRAINER"].get("LR_SCHEDULER", "none")

    def sched_state(epoch_):  # torch _LRScheduler.state_dict() entries PL restores (utils/__init__.py:45-61)
        if sched_kind in (None, "none"):
            return None
        cur = C.lr_at_epoch(cfg, lr, epoch_, args.num_epochs)
        return {"last_epoch": int(epoch_), "_step_count": int(epoch_) + 1, "_last_lr": [cur], "base_lrs": [lr], "kind": sched_kind}

    # bound before the loop: a checkpoint whose epoch is already >= --num_epochs (or a loop whose body never runs) still
    # writes last.ckpt below
    epoch, in_epoch, gen_state0 = first_epoch, skip, gen.get_state()
    for epoch in range(first_epoch, args.num_epochs):
        if hasattr(step_fn.opt, "lr"):  # utils/__init__.py:45-61: the scheduler steps once per epoch
            step_fn.opt.lr = C.lr_at_epoch(cfg, lr, epoch, args.num_epochs)
        gen_state0 = gen.get_state()  # the permutation of this epoch is drawn from here
        in_epoch = skip if epoch == first_epoch else 0
        for b in cache.epoch(args.batch_size, generator=gen, drop_last=True, max_batches=steps_per_epoch, skip_batches=in_epoch):
            rdr.nerf_far_override = False  # neuconw_system.py:343: training always reads near / far from the cache
            ratio = 1.0 if n["ANNEAL_END"] == 0 else min(1.0, step / n["ANNEAL_END"])
            loss, out = step_fn(b["rays"], b["ts"], b["semantics"], b["rgbs"], background_rgb=bg, cos_anneal_ratio=ratio)
            if update_freq > 0 and (step + 1) % update_freq == 0:  # :361-365
                voxel.octree_update(rdr, train_level, n["SDF_THRESHOLD"])
            in_epoch += 1
            if step % cfg["TRAINER"]["SAVE_FREQ"] == 0 and rank == 0:  # :367-374  (written AFTER the step: global_step + 1 done)
                os.makedirs(save_dir, exist_ok=True)
                trainer.save_checkpoint(os.path.join(save_dir, "iter_%d.ckpt" % step), emb, neuconw, nerf,
                                        optimizer=step_fn.opt, global_step=step + 1, epoch=epoch, lr_scheduler=sched_state(epoch),
                                        extra={"ncw_resume": trainer.resume_state(step_fn.opt, rdr, epoch, in_epoch, gen_state0)})
            if rank == 0 and step % args.log_every == 0:
                now = time.perf_counter()
                gn = float(getattr(step_fn, "last_grad_norm", float("nan")))
                print("epoch %d step %d  loss %.5f  s_val %.5f  |grad| %.4g  (%.1f ms/step)%s"
                      % (epoch, step, float(loss), float(out["s_val"]), gn, 1e3 * (now - t_last) / max(1, args.log_every),
                         "" if gn == gn and abs(gn) != float("inf") else
                         "  <- non-finite gradient norm: this update was skipped and the fp16 loss scale halved (now %g; "
                         "%d skipped so far)" % (rdr.grad_scale, step_fn.opt.skipped_steps)))
                t_last = now
            step += 1
            if args.max_steps and step >= args.max_steps:
                done = True
                break
        if done:
            break
    torch.cuda.synchronize()
    if rank == 0:
        os.makedirs(save_dir, exist_ok=True)
        trainer.save_checkpoint(os.path.join(save_dir, "last.ckpt"), emb, neuconw, nerf, optimizer=step_fn.opt, global_step=step,
                                epoch=epoch, lr_scheduler=sched_state(epoch),
                                extra={"ncw_resume": trainer.resume_state(step_fn.opt, rdr, epoch, in_epoch, gen_state0)})
        print("%d steps in %.1f s; wrote %s" % (step, time.perf_counter() - t0, os.path.join(save_dir, "last.ckpt")))
    print("[rank %d] finished after %d steps with %d rays resident" % (rank, step, len(cache)))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
