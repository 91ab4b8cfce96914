#!/usr/bin/env python
"""bench.py -- ray-samples/sec of a full NeuralRecon-W train step on MI355X.

    python bench.py --gpus N --steps K --warmup W         (N>1: launched by torch.distributed.run)

A "step" = NeuconWRenderer.render (sampler + background NeRF + SDF/colour nets + compositor)
+ NeuconWLoss + backward (incl. the second-order SDF terms and every weight gradient) + the
gradient all-reduce + grad-norm clip + Adam step, on a synthetic batch of BASELINE.json's
configs[1]: 1024 rays/GPU x (64 coarse + 64 fine) samples, SDF 8x256, colour 4x256, background
NeRF 8x256, 4 outside samples, bf16 MFMA with f32 accumulation.  Rays shard across ranks (weak
scaling); value = total ray-samples of all ranks / max-over-ranks time.

Prints ONE JSON line (rank 0) with the driver contract keys plus `roofline` (dominant kernel,
algorithmic FLOPs / HIP-event duration) and `cpu_baseline` (the CPU oracle timed on this box).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

W_SDF = 256
R_PER_GPU = 1024
N_SAMPLES, N_IMPORTANCE, UP_STEPS, N_OUTSIDE, S_VAL_BASE = 64, 64, 2, 4, 3
N_A, N_VOCAB = 48, 5000
# algorithmic MACs per point (SURVEY.md 8d / BASELINE.md 4)
M_SDF, M_SDF1, M_COL, M_BG = 524544, 459008, 355968, 659456
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak, MI355X_MICROARCH.md


def synth_batch(R, seed, device):
    g = torch.Generator().manual_seed(seed)
    o = torch.tensor([0.0, 0.0, -2.0]) + 0.1 * torch.randn(R, 3, generator=g)
    d = torch.tensor([0.0, 0.0, 1.0]) + 0.1 * torch.randn(R, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    near, far = torch.full((R, 1), 1.0), torch.full((R, 1), 3.0)
    depth_gt = torch.full((R, 1), 2.0)
    depth_w = (torch.rand(R, 1, generator=g) < 0.2).float()
    rays = torch.cat([o, d, near, far, depth_gt, depth_w], -1)
    ts = torch.randint(0, N_VOCAB, (R,), generator=g)
    label = torch.where(torch.rand(R, generator=g) < 0.1, torch.tensor(2), torch.tensor(0))
    rgbs = torch.rand(R, 3, generator=g)
    return rays.to(device), ts.to(device), label.to(device), rgbs.to(device)


def build_models(device, prec, seed=0):
    import neuralrecon_w_amd as nw

    torch.manual_seed(seed)
    sdf_cfg = dict(d_in=3, d_out=W_SDF + 1, d_hidden=W_SDF, n_layers=8, skip_in=(4,), multires=6, bias=0.5, scale=1,
                   geometric_init=True, weight_norm=True, inside_outside=False)
    color_cfg = dict(d_in=9, d_feature=W_SDF, mode="idr", d_out=3, d_hidden=256, n_layers=4, head_channels=128,
                     static_head_layers=2, weight_norm=True, multires_view=4)
    emb = torch.nn.Embedding(N_VOCAB, N_A)
    neuconw = nw.NeuconW(sdfNet_config=sdf_cfg, colorNet_config=color_cfg, SNet_config=dict(init_val=0.3),
                         in_channels_a=N_A, encode_a=True)
    nerf = nw.NeRF(D=8, d_in=4, d_in_view=3, W=256, multires=10, multires_view=4, output_ch=4, skips=[4],
                   encode_appearance=True, in_channels_a=N_A, in_channels_dir=27, use_viewdirs=True)
    with torch.no_grad():  # exercise weight-norm
        for n, p in neuconw.named_parameters():
            if n.endswith("weight_g"):
                p.mul_(1.0 + 0.1 * torch.randn_like(p))
    emb, neuconw, nerf = emb.to(device), neuconw.to(device), nerf.to(device)
    rdr = nw.NeuconWRenderer(
        nerf=nerf, neuconw=neuconw, embeddings={"a": emb}, n_samples=N_SAMPLES, n_importance=N_IMPORTANCE,
        n_outside=N_OUTSIDE, up_sample_steps=UP_STEPS, perturb=1.0, origin=[0, 0, 0], radius=1.0,
        s_val_base=S_VAL_BASE, spc_options={"recontruct_path": "/nonexistent", "voxel_size": 0.1, "min_track_length": 1},
        sample_range=16, boundary_samples=0, nerf_far_override=False, render_bg=True, trim_sphere=True,
        mesh_mask_list=["sky"], depth_loss=True, prec=prec)
    rdr.sync_free = True  # identical loss value/gradients, no mid-step device->host sync (see renderer.py)
    return emb, neuconw, nerf, rdr


def loss_fn(out, rgbs):
    """NeuconWLoss, losses.py:21-43 with the brandenburg_gate weights (igr 1e-4, mask 0.1, depth 0.1)."""
    R = rgbs.shape[0]
    loss = (out["color"] - rgbs).abs().sum() / (R + 1e-5)
    loss = loss + 1e-4 * out["gradient_error"].mean()
    loss = loss + 0.1 * out["mask_error"].mean()
    loss = loss + 0.1 * out["sfm_depth_loss"].mean()
    return loss


def kernel_flops(R):
    """Algorithmic FLOPs per launch of each C-ABI kernel at the bench shape (1 MAC = 2 FLOP)."""
    S, O = N_SAMPLES + N_IMPORTANCE, N_OUTSIDE
    n_in, n_bg = R * S, R * (S + O)
    per_step_imp = N_IMPORTANCE // UP_STEPS
    # FLOPs per STEP of each entry point (summed over its launches in one step)
    return {
        "ncw_sdf_fwd": 2.0 * n_in * 2 * M_SDF,             # forward + input-adjoint
        "ncw_sdf_bwd": 2.0 * n_in * 2 * M_SDF,             # W qbar and W^T zbar chains
        "ncw_color_fwd": 2.0 * n_in * M_COL,
        "ncw_color_bwd": 2.0 * n_in * M_COL,
        "ncw_nerf_fwd": 2.0 * n_bg * M_BG,
        "ncw_nerf_bwd": 2.0 * n_bg * M_BG,
        # weight-gradient GEMMs of all three networks: 2 batches (inside + background) x 2 tile variants
        "ncw_wgrad_tiled": 2.0 * n_in * (2 * M_SDF + M_COL) + 2.0 * n_bg * M_BG,
        "ncw_wgrad": 2.0 * n_in * (2 * M_SDF + M_COL) + 2.0 * n_bg * M_BG,   # f32 mode uses this entry point
        "ncw_sdf_infer_rays": 2.0 * R * (N_SAMPLES + per_step_imp * (UP_STEPS - 1)) * M_SDF1,
    }


def cpu_baseline(sample_rays=32, repeats=2, max_threads=32):
    """The CPU oracle (oracle/neuconw_oracle.py, pinned to the real reference by tests/golden) timed on
    this box's host cores on a bounded sample of the same workload (same nets, same sampler shape)."""
    from oracle import neuconw_oracle as O

    # torch's intra-op pool stops scaling (and thrashes) far below a 256-core host's core count at this
    # problem size: use at most `max_threads` threads and report exactly that number as `cores`.
    cores = min(os.cpu_count() or 1, max_threads)
    torch.set_num_threads(cores)
    emb, neuconw, nerf, _ = build_models("cpu", 0)
    sd = {"embedding_a.weight": emb.weight.detach()}
    sd.update({"neuconw." + k: v.detach() for k, v in neuconw.state_dict().items() if not k.startswith("xyz_enc")})
    sd.update({"nerf." + k: v.detach() for k, v in nerf.state_dict().items()})
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    cfg = dict(n_samples=N_SAMPLES, n_importance=N_IMPORTANCE, n_outside=N_OUTSIDE, up_sample_steps=UP_STEPS,
               s_val_base=S_VAL_BASE, render_bg=True, trim_sphere=True, mesh_mask_list=["sky"], depth_loss=True,
               igr_weight=1e-4, mask_weight=0.1, depth_weight=0.1, skip_in=(4,), multires=6, multires_view=4)
    rays, ts, label, rgbs = synth_batch(sample_rays, 123, "cpu")
    times = []
    for i in range(repeats + 1):
        t0 = time.perf_counter()
        out = O.render(sd, cfg, rays, ts, label, 0.5, torch.zeros(1, 3))
        loss = O.neuconw_loss(out, rgbs, cfg)
        torch.autograd.grad(loss, [v for v in sd.values() if v.requires_grad], allow_unused=True)
        times.append(time.perf_counter() - t0)
    t = sorted(times[1:])[len(times[1:]) // 2]
    S = N_SAMPLES + N_IMPORTANCE
    return {"value": sample_rays * S / t, "unit": "ray-samples/s", "cores": cores, "kind": "port",
            "sample": "%d rays x %d samples, same networks/sampler, fp32 torch-CPU oracle, render+loss+backward, "
                      "median of %d after 1 warm-up (%.2f s/step)" % (sample_rays, S, repeats, t)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--prec", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--rays", type=int, default=R_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config", default="headline", choices=["headline", "shipped"],
                    help="headline = BASELINE.json configs[1] (the metric's shape); shipped = the reference's yaml shape "
                         "(W=512 SDF, 8+16 samples): a secondary row, never the reported metric")
    ap.add_argument("--graph", action="store_true",
                    help="record the step into HIP graphs and replay it (trainer.TrainStep(capture=True)); measured "
                         "4.78 vs 4.80 ms eager on one MI355X -- the step is not host-launch-bound -- so eager is the default")
    ap.add_argument("--dist-check", action="store_true",
                    help="initialise the process group, print {world, ranks, devices} from rank 0 and exit (launcher test; "
                         "needs no GPU with NCW_DIST_BACKEND=gloo)")
    args = ap.parse_args()
    # ---- self-launch: `python bench.py --gpus N` (N > 1) outside torch.distributed.run starts the N ranks itself,
    # one process per GPU over RCCL (train.py:53-55: gpus=N, accelerator='ddp'); under torchrun WORLD_SIZE is set
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket

        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execvp(sys.executable, cmd)
    if args.config == "shipped":  # config/train_brandenburg_gate.yaml: SDF 8x512, N_SAMPLES 8, N_IMPORTANCE 16
        globals().update(W_SDF=512, N_SAMPLES=8, N_IMPORTANCE=16, M_SDF=2097664, M_SDF1=1835520, M_COL=585344)

    import neuralrecon_w_amd as nw
    from neuralrecon_w_amd import ddp
    from neuralrecon_w_amd import lib as L

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if os.environ.get("NCW_BENCH_ONE_GPU_TEST"):  # plumbing test: N ranks share GPU 0 over gloo
            local_rank = 0
        backend = os.environ.get("NCW_DIST_BACKEND", "nccl")
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (torch.distributed.run --nproc-per-node must equal --gpus)"
                         % (args.gpus, world))
    rank_info = {"rank": rank, "local_rank": local_rank, "pid": os.getpid(),
                 "HIP_VISIBLE_DEVICES": os.environ.get("HIP_VISIBLE_DEVICES"),
                 "device": (torch.cuda.get_device_name(local_rank) if torch.cuda.is_available() else None)}
    ranks = [rank_info]
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, rank_info)
    if args.dist_check:
        if rank == 0:
            print(json.dumps({"dist_check": True, "world": dist.get_world_size() if world > 1 else 1,
                              "backend": dist.get_backend() if world > 1 else None, "ranks": ranks}))
        if world > 1:
            dist.destroy_process_group()
        return
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    prec = nw.PREC_BF16 if args.prec == "bf16" else nw.PREC_F32
    emb, neuconw, nerf, rdr = build_models(dev, prec)
    # LR rule of train.py:21-25: 1e-4 * world*batch / 4096; Adam eps 1e-7 (utils/__init__.py:24-31); clip 0.99
    # (train.py:61).  TrainStep = render + loss + backward + one flat all-reduce + clip + Adam (trainer.py).
    R = args.rays
    # --graph: the step is recorded once into HIP graphs and replayed (trainer.py): same kernels, same arithmetic,
    # one graph launch instead of ~130 launches; with N > 1 the RCCL all-reduce stays eager between two graphs.
    train = nw.TrainStep(rdr, [emb, neuconw, nerf], loss_fn, lr=1e-4 * world * R / 4096.0, eps=1e-7, clip=0.99,
                         world_size=world, capture=args.graph, capture_warmup=3)
    rays, ts, label, rgbs = synth_batch(R, 1000 + rank, dev)
    bg = torch.zeros(1, 3, device=dev)

    def step(i):
        loss, _ = train(rays, ts, label, rgbs, background_rgb=bg, cos_anneal_ratio=min(1.0, i / 50000.0))
        return loss

    if args.graph:  # setup, not warm-up: 3 eager steps + the capture happen before the W warm-up steps
        for i in range(4):
            step(0)
    for i in range(args.warmup):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    S = N_SAMPLES + N_IMPORTANCE
    value = world * R * S * args.steps / dt

    # ---- per-kernel HIP-event timing over a few more live steps (same stream) -> roofline ---------------
    roofline = None
    # every rank runs the same extra steps (the step contains the gradient all-reduce); only rank 0 records
    prof_steps = max(2, min(5, args.steps))
    if rank == 0:
        L.PROFILE = {}
    for i in range(prof_steps):
        train.eager_step(rays, ts, label, rgbs, background_rgb=bg,
                         cos_anneal_ratio=min(1.0, (args.warmup + args.steps + i) / 50000.0))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    if rank == 0:
        prof, L.PROFILE = L.PROFILE, None
        fl = kernel_flops(R)
        rows = {}
        for name, evs in prof.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            rows[name] = (sum(ms) / prof_steps, len(evs) // prof_steps, ms)
        total_ms = sum(v[0] for v in rows.values())
        dom = max((k for k in rows if fl.get(k) is not None), key=lambda k: rows[k][0])
        avg_ms = rows[dom][0] / rows[dom][1]
        ach = fl[dom] / (rows[dom][0] * 1e-3) / 1e12
        peak = PEAK_BF16_TFLOPS if prec == nw.PREC_BF16 else 157.3
        # HBM traffic of the dominant entry point per launch: rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, KB)
        # of this same command, committed under profiles/ (bench.py cannot run rocprof on itself).
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01", "pmc_traffic_v3.json")))
            keys = {"ncw_wgrad_tiled": "wgrad_dma_kernel", "ncw_sdf_bwd": "sdf_bwd_kernel", "ncw_sdf_fwd": "sdf_fwd_kernel"}
            sel = [v for k, v in tj.items() if keys.get(dom, "\0") in k]
            if sel and prec == nw.PREC_BF16 and R == R_PER_GPU:
                # MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE counts wide coalesced streaming reads (16 B/lane,
                # global_load and LDS-DMA alike) at exactly 1/2 -> doubled; WRITE_SIZE is taken as reported.  KB.
                kb = sum((2.0 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * v["launches"] for v in sel)
                traffic = round(kb * 1024.0 / sum(v["launches"] for v in sel), 0)  # bytes per launch
        except Exception:
            traffic = None
        roofline = {"kernel": dom, "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": traffic, "avg_launch_ms": round(avg_ms, 4),
                    "launches_per_step": rows[dom][1],
                    "algorithmic_gflop_per_launch": round(fl[dom] / rows[dom][1] / 1e9, 2),
                    "kernel_tflops": {k: round(fl[k] / (rows[k][0] * 1e-3) / 1e12, 1) for k in rows if fl.get(k)},
                    "per_step_kernel_ms": {k: round(v[0], 4) for k, v in sorted(rows.items(), key=lambda kv: -kv[1][0])},
                    "sum_kernel_ms_per_step": round(total_ms, 4)}
        if dom.startswith("ncw_wgrad"):
            # The weight-gradient GEMMs reduce over the POINTS: every product streams its two stash operands
            # once (algorithmic bytes = sum over products of (rbx + rby) x 32 features x elem x points); at
            # ~180 FLOP/B they sit left of the ridge (2.5 PFLOP/s / 8 TB/s = 312 FLOP/B): HBM-bound.
            esz = 2 if prec == nw.PREC_BF16 else 4
            alg_bytes = 0.0
            for ent in neuconw.sdf_net.__dict__.get("_stash_cache")._e.values():
                wb = ent.get("wgrad_batch")
                if wb is not None:
                    alg_bytes += type(wb).algorithmic_bytes(wb.items, esz)
            gbs = alg_bytes / (rows[dom][0] * 1e-3) / 1e9
            roofline.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s",
                             "frac": round(gbs / 8000.0, 4), "algorithmic_gbytes_per_step": round(alg_bytes / 1e9, 3),
                             "mfma_tflops": round(ach, 1)})
        # step-level MFMA fraction: all algorithmic FLOPs of the step / wall time
        step_flops = 2.0 * R * ((N_SAMPLES + (UP_STEPS - 1) * N_IMPORTANCE // UP_STEPS) * M_SDF1
                                + S * (6 * M_SDF + 3 * M_COL) + (S + N_OUTSIDE) * 3 * M_BG)
        roofline["step_algorithmic_tflop"] = round(step_flops / 1e12, 4)
        roofline["step_frac_of_mfma_peak"] = round(step_flops / (dt / args.steps) / 1e12 / peak, 4)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline()
        except Exception as e:  # the baseline must never take the bench line down
            cpu = {"value": None, "unit": "ray-samples/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (e,)}

    if rank == 0:
        line = {
            "metric": ("ray-samples/sec (train step) at 1024 rays x 128 samples" if args.config == "headline" else
                       "ray-samples/sec (train step) at %d rays x %d samples [secondary shape]" % (R, S)), "value": value,
            "unit": "ray-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.prec, "data": "synthetic",
            "config": {"workload": "brandenburg_gate config (%s): %d rays/GPU x (%d coarse + %d fine) "
                                   "samples, SDF 8x%d + colour 4x256 + bg NeRF 8x256, 4 outside samples, up_sample_steps 2, "
                                   "render+loss+backward+allreduce+clip+Adam"
                                   % ("BASELINE.json configs[1]" if args.config == "headline" else "shipped yaml shape, secondary",
                                      R, N_SAMPLES, N_IMPORTANCE, W_SDF),
                       "rays_per_gpu": R, "samples_per_ray": S, "global_rays": world * R, "parallelism": "dp%d" % world,
                       "world_size": world, "ranks": ranks,
                       "submission": "hip-graph replay" if args.graph else "eager",
                       "final_loss": float(loss.detach())},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
