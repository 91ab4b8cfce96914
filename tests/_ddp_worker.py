"""Worker of tests/test_gpu_ddp.py: one rank of a world-size-N data-parallel job whose ranks SHARE GPU 0 (the builder's lease
is one GPU; gloo carries the collectives, staging the device buffers through the host).  Each rank runs the HIP
`TrainStep` on its own shard of the ray batch (rank r owns rays [r * R, (r + 1) * R), train.py:53-55).

Checks, per precision: (1) after every step the replicas' flat parameter buffers are bit-identical; (2) the averaged flat
gradient of step 1 equals the mean over ranks of the fp64 ORACLE's gradient evaluated per shard (gradient_error is a
per-rank-global scalar, rendering/renderer.py:757-765: the oracle is evaluated per shard, not on the concatenated batch).
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
    from oracle import neuconw_oracle as O
    from tests._build import build_system, named_params, state_dict_cpu
    from tests._parity import CFG, perturb_weights
    from tests._util import synth_rays

    R, W, ns, ni, steps = int(os.environ.get("NCW_DDP_R", "48")), int(os.environ.get("NCW_DDP_W", "64")), 16, 16, 3
    results = {}
    for pname in os.environ.get("NCW_DDP_PRECS", "f32,f16").split(","):
        prec = {"f32": nw.PREC_F32, "f16": nw.PREC_F16, "bf16": nw.PREC_BF16}[pname]
        big = W >= 256
        emb, neuconw, nerf, rdr = build_system(W=W, n_a=48 if big else 16, n_vocab=100, nerf_w=256 if big else 64,
                                               color_hidden=256 if big else 64, head=128 if big else 32,
                                               seed=5 + rank,  # ranks start DIFFERENT: TrainStep's broadcast must fix it
                                               prec=prec, n_samples=ns, n_importance=ni)
        perturb_weights(neuconw, 0.1, 0.02, seed=11 + rank)
        rdr.sync_free = True
        loss_fn = nw.NeuconWLoss(coef=1.0, igr_weight=0.1, mask_weight=0.1, depth_weight=0.1, use_mask=True, use_depth=True)
        ts_ = nw.TrainStep(rdr, [emb, neuconw, nerf], loss_fn, lr=1e-3, eps=1e-7, clip=0.99, world_size=world)
        rays_all, t_all, label_all, rgbs_all = synth_rays(R * world, 91, 100)
        sl = slice(rank * R, (rank + 1) * R)
        rays, tt, label, rgbs = rays_all[sl].cuda(), t_all[sl].cuda(), label_all[sl].cuda(), rgbs_all[sl].cuda()
        bg = torch.zeros(1, 3).cuda()
        # ---- oracle gradient of THIS shard at the (broadcast) initial weights -------------------------------------
        sd = state_dict_cpu(emb, neuconw, nerf, torch.float64)
        sd = {k: v.requires_grad_(True) for k, v in sd.items()}
        cfg = dict(CFG, n_samples=ns, n_importance=ni)
        ref = O.render(sd, cfg, rays_all[sl].double(), t_all[sl], label_all[sl], 0.3, torch.zeros(1, 3, dtype=torch.float64))
        lref = O.neuconw_loss(ref, rgbs_all[sl].double(), cfg)
        names = list(sd)
        gl = torch.autograd.grad(lref, [sd[k] for k in names], allow_unused=True)
        gref = {k: (torch.zeros_like(sd[k]) if g is None else g) for k, g in zip(names, gl)}
        for k in names:  # mean over ranks of the per-shard oracle gradients
            dist.all_reduce(gref[k])
            gref[k] /= world
        # ---- step 1, split into its halves so the averaged gradient can be looked at before clip + Adam ------------
        params = named_params(emb, neuconw, nerf)
        loss, _ = ts_._fwd_bwd(rays, tt, label, rgbs, bg, 0.3, dict(perturb_overwrite=0))
        h = ts_.fp.allreduce(world, None, async_op=True)
        h.wait()
        torch.cuda.synchronize()

        def net_of(k):
            return k.split(".")[0] if not k.startswith("neuconw.") else ".".join(k.split(".")[:2])

        scale, worst, worst_k = {}, 0.0, None
        for k, g in gref.items():
            scale[net_of(k)] = max(scale.get(net_of(k), 0.0), float(g.abs().max()))
        for k, g in gref.items():
            if k not in params or params[k].grad is None:
                continue
            e = float((params[k].grad.detach().cpu().double() - g).abs().max()) / max(scale[net_of(k)], 1e-30)
            if e > worst:
                worst, worst_k = e, k
        ts_._update()
        sums, same = [], True
        for i in range(steps):
            if i > 0:
                ts_(rays, tt, label, rgbs, background_rgb=bg, cos_anneal_ratio=0.3, perturb_overwrite=0)
            torch.cuda.synchronize()
            flat = ts_.fp.flat.detach().cpu()
            ref0 = flat.clone()
            dist.broadcast(ref0, src=0)
            same = same and bool(torch.equal(ref0, flat))
            sums.append(float(flat.double().abs().sum()))
        flags = [None] * world
        dist.all_gather_object(flags, (same, bool(torch.isfinite(flat).all())))
        results[pname] = dict(replicas_identical=all(f[0] for f in flags), finite=all(f[1] for f in flags),
                              grad_err_vs_shard_oracle_mean=worst, worst_param=worst_k, loss_rank0=float(loss.detach()),
                              param_abs_sums=sums, applied_steps=ts_.opt.step_count, skipped=ts_.opt.skipped_steps)
        del ts_, rdr, emb, neuconw, nerf
    if rank == 0:
        print(json.dumps({"ddp_worker": True, "world": world, "rays_per_rank": R, "W": W, "results": results}))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
