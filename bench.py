#!/usr/bin/env python
"""bench.py -- ray-samples/sec of a full NeuralRecon-W train step on MI355X.

    python bench.py --gpus N --steps K --warmup W         (N>1: launched by torch.distributed.run)

A "step" = NeuconWRenderer.render (sampler + background NeRF + SDF/colour nets + compositor)
+ NeuconWLoss + backward (incl. the second-order SDF terms and every weight gradient) + the
gradient all-reduce + grad-norm clip + Adam step, on a synthetic batch of BASELINE.json's
configs[1]: 1024 rays/GPU x (64 coarse + 64 fine) samples, SDF 8x256, colour 4x256, background
NeRF 8x256, 4 outside samples, fp16 MFMA operands with f32 accumulation, a dynamic loss scale, and the SDF VALUE chain in
split precision (fp16 hi + lo pairs, three MFMAs per product: fp32-like SDF values -- csrc/ncw_split.hip), as are the adjoint
sweep's weights, the colour network's weights and point / normal inputs and -- at the samples the compositor can use -- the whole
background NeRF (ncw_nerf_refine) (--prec f16, the
default: rendered outputs within 1e-4 of the oracle at initialisation AND at trained sharpness, `parity`; --prec bf16 | f32
select the others, `plain_f16_mode` times the step without the split).  Rays
shard across ranks (weak scaling); value = total ray-samples of all ranks / max-over-ranks time.

Prints ONE JSON line (rank 0) with the driver contract keys plus
  `roofline`     dominant kernel by HIP-event time: `frac_mfma` (algorithmic FLOPs / duration / 2.5 PFLOP/s)
                 AND `frac_hbm` (algorithmic stash bytes / duration / 8 TB/s) -- `frac` is the one for `bound`; `traffic`
                 = HBM bytes per launch MEASURED WITH THIS RUN (two rocprofv3 --pmc passes of a short re-run of this
                 script: FETCH_SIZE x 2 per MI355X_MICROARCH.md + WRITE_SIZE), `step_traffic_gb` = all kernels of a step;
  `parity_mode`  the same step in the fp32 parity mode (the <= 1e-4 mode), a few steps timed in the same process;
  `alt_mode`     the same step in the other 16-bit type (bf16 when --prec f16), timed in the same process;
  `bg_elimination` the same step as the PRODUCT runs it by default: the background NeRF only where the compositor can use
                 its output (identical results; `value` itself evaluates every sample like the reference);
  `cpu_baseline` the CPU oracle on the first 256 rays of the SAME batch, timed on this box;
  `parity`       the measured error of the timed program: GPU render + loss of those 256 rays in the timed dtype vs the
                 oracle, at the initial operating point and at inv_s = 403 (where NeuS trains), fp32 mode beside it.
Secondary rows (never the reported metric): --config shipped | voxel (BASELINE configs[2]) | grid512 (configs[4]).
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
N_BOUNDARY = 0  # 10 in --config voxel (boundary samples of the fine-octree window)
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


def loss_fn_torch(out, rgbs):
    """NeuconWLoss, losses.py:21-43 with the brandenburg_gate weights (igr 1e-4, mask 0.1, depth 0.1), plain torch --
    the arithmetic `loss_fn` below runs as one launch each way (tests/test_gpu_glue.py compares them)."""
    R = rgbs.shape[0]
    loss = (out["color"] - rgbs).abs().sum() / (R + 1e-5)
    loss = loss + 1e-4 * out["gradient_error"].mean()
    loss = loss + 0.1 * out["mask_error"].mean()
    loss = loss + 0.1 * out["sfm_depth_loss"].mean()
    return loss


def loss_fn(out, rgbs):
    """The same loss through neuralrecon_w_amd.NeuconWLoss (`ncw_loss_fwd` / `ncw_loss_bwd`)."""
    global _LOSS
    if _LOSS is None:
        import neuralrecon_w_amd as nw

        _LOSS = nw.NeuconWLoss(coef=1.0, igr_weight=1e-4, mask_weight=0.1, depth_weight=0.1, use_mask=True, use_depth=True)
    return _LOSS(out, rgbs)


_LOSS = None


def kernel_flops(R):
    """Algorithmic FLOPs per launch of each C-ABI kernel at the bench shape (1 MAC = 2 FLOP)."""
    S, O = N_SAMPLES + N_IMPORTANCE + N_BOUNDARY, N_OUTSIDE
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


ORACLE_CFG = None
BATCH_SEED = 1000  # rank r times synth_batch(R, BATCH_SEED + r); the parity / cpu_baseline legs use rank 0's batch (--seed: robustness sweeps)


def _oracle_setup(sample_rays, seed, variance=None):
    """The oracle's inputs for the first `sample_rays` rays of rank 0's timed batch: the same networks (same seed, same
    initial weights as build_models on the GPU), optionally with SingleVarianceNetwork.variance overridden."""
    emb, neuconw, nerf, _ = build_models("cpu", 0)
    sd = {"embedding_a.weight": emb.weight.detach()}
    sd.update({"neuconw." + k: v.detach() for k, v in neuconw.state_dict().items() if not k.startswith("xyz_enc")})
    sd.update({"nerf." + k: v.detach() for k, v in nerf.state_dict().items()})
    if variance is not None:
        sd["neuconw.deviation_network.variance"] = torch.tensor(float(variance))
    cfg = dict(n_samples=N_SAMPLES, n_importance=N_IMPORTANCE, n_outside=N_OUTSIDE, up_sample_steps=UP_STEPS,
               s_val_base=S_VAL_BASE, render_bg=True, trim_sphere=True, mesh_mask_list=["sky"], depth_loss=True,
               igr_weight=1e-4, mask_weight=0.1, depth_weight=0.1, skip_in=(4,), multires=6, multires_view=4)
    batch = [t[:sample_rays] for t in synth_batch(R_PER_GPU, seed, "cpu")]  # rank 0's batch, first rays
    return sd, cfg, batch


# The timed CPU program is the oracle (kind "port"): the GPU box has no /root/reference.  Its speed relative to the UNMODIFIED
# reference (rendering/renderer.py render + NeuconWLoss + backward through the double forward + autograd.grad, ~9 M_sdf per
# sample against the oracle's 6) was calibrated once where both exist (this repo's build container, 8 host threads, the same
# 256 rays): scripts/diag/port_over_reference.py -> profiles/r04/port_over_reference.json.  None until measured.
RECORDED_FILE = os.path.join(ROOT, "profiles", "r04", "port_over_reference.json")


def recorded_calibration():
    """Numbers that can only be measured where /root/reference exists (the build container): read at run time from the committed
    file scripts/diag/port_over_reference.py wrote -- never constants in this script -- and tagged as RECORDED, not measured in
    this run.  -> (port_over_reference dict or None, reference-fp32-vs-fp64 dict or None)."""
    try:
        with open(RECORDED_FILE) as fh:
            d = json.load(fh)
        t = d["timing"]
        por = {"value": round(t["port_over_reference"], 4), "recorded_not_measured_in_this_run": True,
               "source": os.path.relpath(RECORDED_FILE, ROOT),
               "measured": "the unmodified reference %.2f s/step vs the oracle %.2f s/step on the same %d rays of this batch"
                           % (t["reference_s_per_step"], t["oracle_s_per_step"], d["rays"]),
               "calibration_threads": d["threads"], "calibration_host_cpu_count": d["cpu_count"],
               "note": "calibrated on ANOTHER box with %d threads; this run's cpu_baseline uses `cores` threads of this box"
                       % d["threads"],
               "meaning": "reference time / port time: the reference's CPU throughput is this factor BELOW cpu_baseline.value"}
        r = d.get("reference_fp32_on_the_timed_batch_256_rays", {})
        keys = ("colour", "depth", "weights_sum", "weights")
        ref = {"recorded_not_measured_in_this_run": True, "source": os.path.relpath(RECORDED_FILE, ROOT)}
        for name, k in (("inv_s_20", "variance_0.3"), ("inv_s_403", "variance_0.6"), ("trained_40_steps_inv_s_403", "trained_40_steps")):
            if k in r:
                ref[name] = {q: float("%.3g" % r[k][q]) for q in keys if q in r[k]}
        return por, (ref if len(ref) > 2 else None)
    except Exception:
        return None, None


def cpu_baseline(sample_rays=256, repeats=3, max_threads=32, seed=None, full_batch=True):
    seed = BATCH_SEED if seed is None else seed
    """The CPU oracle (oracle/neuconw_oracle.py, pinned to the real reference by tests/golden) timed on
    this box's host cores on a bounded sample of the same workload (same nets, same sampler shape).
    Returns (cpu_baseline dict, reference outputs of that sample for the `parity` object)."""
    from oracle import neuconw_oracle as O

    # torch's intra-op pool stops scaling (and thrashes) far below a 256-core host's core count at this
    # problem size: use at most `max_threads` threads and report exactly that number as `cores`.
    cores = min(os.cpu_count() or 1, max_threads)
    torch.set_num_threads(cores)
    sd, cfg, (rays, ts, label, rgbs) = _oracle_setup(sample_rays, seed)
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    times = []
    for i in range(repeats + 1):
        t0 = time.perf_counter()
        out = O.render(sd, cfg, rays, ts, label, 0.5, torch.zeros(1, 3))
        loss = O.neuconw_loss(out, rgbs, cfg)
        torch.autograd.grad(loss, [v for v in sd.values() if v.requires_grad], allow_unused=True)
        times.append(time.perf_counter() - t0)
    t = sorted(times[1:])[len(times[1:]) // 2]
    S = N_SAMPLES + N_IMPORTANCE
    ref = {k: out[k].detach() for k in ("color", "depth", "weights_sum", "weights", "z_vals", "gradients", "cdf_fine")}
    ref["loss"] = float(loss.detach())
    ref["sdf"], ref["pts"] = _oracle_sdf_at_samples(sd, rays, ref["z_vals"])
    # BASELINE.md 3 quotes the CPU path on the FULL 1024-ray batch: one un-warmed pass over all of it (the 256-ray sample above is
    # the repeated, median-of-3 figure; per ray-sample the CPU is slower at 1024 rays -- BASELINE.md 2: 10.1 k vs 14.1 k -- so
    # the sample flatters the CPU side).  Skipped when the sample says it would take more than ~40 s.
    full = None
    if full_batch and t * (R_PER_GPU / sample_rays) < 40.0:
        try:
            del out, loss
            sd_f, cfg_f, (rays_f, ts_f, label_f, rgbs_f) = _oracle_setup(R_PER_GPU, seed)
            sd_f = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd_f.items()}
            t0 = time.perf_counter()
            out_f = O.render(sd_f, cfg_f, rays_f, ts_f, label_f, 0.5, torch.zeros(1, 3))
            loss_f = O.neuconw_loss(out_f, rgbs_f, cfg_f)
            torch.autograd.grad(loss_f, [v for v in sd_f.values() if v.requires_grad], allow_unused=True)
            tf = time.perf_counter() - t0
            full = {"value": R_PER_GPU * S / tf, "unit": "ray-samples/s", "seconds": round(tf, 2), "rays": R_PER_GPU, "passes": 1,
                    "note": "all %d rays of the timed batch, ONE pass, no warm-up (BASELINE.md 3)" % R_PER_GPU}
            del out_f, loss_f, sd_f
        except Exception as e:  # (host memory: ~10 GB of autograd state at 1024 rays)
            full = {"value": None, "error": "%r" % (e,)}
    return {"value": sample_rays * S / t, "unit": "ray-samples/s", "cores": cores, "kind": "port", "repeats": repeats,
            "full_batch": full,
            "port_over_reference": recorded_calibration()[0],
            "sample": "the first %d rays of the timed %d-ray batch x %d samples (BASELINE.md 3), same networks / sampler, "
                      "fp32 torch-CPU oracle (the analytic-adjoint restatement: 6 M_sdf per sample where the reference's "
                      "double forward + autograd.grad spends ~9 M_sdf, SURVEY 8d -- this flatters the CPU side slightly), "
                      "render+loss+backward, median of %d after 1 warm-up (%.2f s/step)"
                      % (sample_rays, R_PER_GPU, S, repeats, t)}, ref


def _oracle_sdf_at_samples(sd, rays, z):
    """The SDF network (north_star names SDF among the outputs; render() itself does not return it) at the oracle's own
    sample positions o + z d of every ray: the points the sampler and the compositor query.  -> (sdf [R*S], pts [R*S,3])."""
    from oracle import neuconw_oracle as O

    with torch.no_grad():
        pts = (rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., None]).reshape(-1, 3)
        sdf, _, _ = O.sdf_net({k[len("neuconw."):]: v.detach() for k, v in sd.items() if k.startswith("neuconw.sdf_net.")},
                              pts.to(z.dtype), "sdf_net.", with_grad=False)
    return sdf, pts


def oracle_outputs(sample_rays=256, seed=None, variance=None, state=None, voxel=False):
    """Forward-only oracle evaluation of the same sample (fp64), optionally at another variance (inv_s = exp(10 variance))
    or with another state_dict (`state`: the trained-weights point); voxel: configs[2] (coarse + fine level-7 shell octree)."""
    from oracle import neuconw_oracle as O

    seed = BATCH_SEED if seed is None else seed
    sd, cfg, (rays, ts, label, rgbs) = _oracle_setup(sample_rays, seed, variance)
    if state is not None:
        sd = dict(state)
    sd = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    octs = {}
    if voxel:
        oct_ = dict(occ=voxel_shell(7, device="cpu"), scene_origin=torch.zeros(3, dtype=torch.float64), scale=1.0, voxel_size=2.0 / 128)
        octs = dict(coarse_octree=oct_, fine_octree=oct_)
        cfg = dict(cfg, boundary_samples=10, sample_range=16, radius=1.0, voxel_size=2.0 / 128)
    with torch.no_grad():
        out = O.render(sd, cfg, rays.double(), ts, label, 0.5, torch.zeros(1, 3, dtype=torch.float64), **octs)
        loss = O.neuconw_loss(out, rgbs.double(), cfg)
    ref = {k: out[k] for k in ("color", "depth", "weights_sum", "weights", "z_vals", "gradients", "cdf_fine")}
    ref["loss"] = float(loss)
    ref["sdf"], ref["pts"] = _oracle_sdf_at_samples(sd, rays.double(), ref["z_vals"])
    return ref


def _state_dict_of(emb, neuconw, nerf):
    sd = {"embedding_a.weight": emb.weight.detach().cpu().clone()}
    sd.update({"neuconw." + k: v.detach().cpu().clone() for k, v in neuconw.state_dict().items() if not k.startswith("xyz_enc")})
    sd.update({"nerf." + k: v.detach().cpu().clone() for k, v in nerf.state_dict().items()})
    return sd


def trained_state(dev, steps=40, sample_rays=256, seed=None, lr=1e-3, variance=0.6):
    """A NON-initial operating point for `parity`: `steps` TrainSteps in the fp32 mode (bitwise reproducible) from the bench's
    initial weights on the parity sample, then SingleVarianceNetwork.variance set to `variance` (inv_s 403, where NeuS
    trains) -- the recipe of tests/test_gpu_fullsize.py::test_train_step_vs_oracle_after_training.  -> CPU state_dict."""
    import neuralrecon_w_amd as nw

    seed = BATCH_SEED if seed is None else seed
    emb, neuconw, nerf, rdr = build_models(dev, nw.PREC_F32)
    train = nw.TrainStep(rdr, [emb, neuconw, nerf], loss_fn, lr=lr, eps=1e-7, clip=0.99)
    rays, ts, label, rgbs = [t[:sample_rays] for t in synth_batch(R_PER_GPU, seed, dev)]
    bg = torch.zeros(1, 3, device=dev)
    for i in range(steps):
        train(rays, ts, label, rgbs, background_rgb=bg, cos_anneal_ratio=0.5, perturb_overwrite=0)
    with torch.no_grad():
        neuconw.deviation_network.variance.fill_(float(variance))
    torch.cuda.synchronize()
    return _state_dict_of(emb, neuconw, nerf)


def voxel_setup(rdr, dev):
    """BASELINE configs[2] on a renderer: level-7 shell occupancy as coarse octree (ray near/far) AND fine octree (+-16-voxel
    sampling window, 10 boundary samples).  -> the oracle's (coarse_octree, fine_octree, cfg additions)."""
    from neuralrecon_w_amd import voxel

    occ = voxel_shell(7, device=dev)
    vs = 2.0 / 128
    rdr.nerf_far_override, rdr.voxel_size = True, vs
    rdr.octree_data = voxel.occupancy_from_dense(occ, torch.zeros(3), 1.0, voxel_size=vs)
    rdr.fine_octree_data = voxel.occupancy_from_dense(occ, torch.zeros(3), 1.0, voxel_size=vs)
    rdr.sample_range, rdr.boundary_samples = 16, 10
    oct_ = dict(occ=occ.cpu(), scene_origin=torch.zeros(3), scale=1.0, voxel_size=vs)
    return oct_, oct_, dict(boundary_samples=10, sample_range=16, radius=1.0, voxel_size=vs)


def gpu_outputs(dev, prec, sample_rays=256, seed=None, variance=None, pts=None, state=None, z_override=None, voxel=False):
    """The product's render + loss of the same sample, same (initial, or `state`) weights, in the TIMED precision,
    deterministic sampling (perturb 0) like the oracle leg; `pts`: where to evaluate the SDF network (the oracle's samples);
    `z_override`: the oracle's own primary sample depths (the MLPs + compositor at FIXED positions); `voxel`: configs[2]."""
    seed = BATCH_SEED if seed is None else seed
    emb, neuconw, nerf, rdr = build_models(dev, prec)
    if voxel:
        voxel_setup(rdr, dev)
    if state is not None:
        with torch.no_grad():
            emb.weight.copy_(state["embedding_a.weight"])
            neuconw.load_state_dict({k[len("neuconw."):]: v for k, v in state.items() if k.startswith("neuconw.")}, strict=False)
            nerf.load_state_dict({k[len("nerf."):]: v for k, v in state.items() if k.startswith("nerf.")})
    if variance is not None:
        with torch.no_grad():
            neuconw.deviation_network.variance.fill_(float(variance))
    rays, ts, label, rgbs = [t[:sample_rays] for t in synth_batch(R_PER_GPU, seed, dev)]
    with torch.no_grad():
        out = rdr.render(rays, ts, label, perturb_overwrite=0, background_rgb=torch.zeros(1, 3, device=dev),
                         cos_anneal_ratio=0.5, _z_override=z_override)
        loss = loss_fn_torch(out, rgbs)
    got = {k: out[k].detach().cpu() for k in ("color", "depth", "weights_sum", "weights", "gradients", "cdf_fine")}
    got["loss"] = float(loss)
    if pts is not None:  # the SDF network in the timed precision (the sampler's / compositor's queries)
        with torch.no_grad():
            got["sdf"] = neuconw.sdf(pts.float().to(dev), prec).reshape(-1).cpu()
    return got


def _rel(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def parity_errors(got, ref):
    """max |gpu - oracle| / max |oracle| per output (the north star's '1e-4 rel' measure) + the loss difference."""
    e = {"colour": _rel(got["color"], ref["color"]), "depth": _rel(got["depth"], ref["depth"]),
         "weights_sum": _rel(got["weights_sum"], ref["weights_sum"]), "loss": abs(got["loss"] - ref["loss"])}
    # how the colour error is distributed over the rays (the headline number is the MAX over rays and channels)
    per_ray = (got["color"].double() - ref["color"].double()).abs().amax(dim=-1) / (ref["color"].double().abs().max() + 1e-12)
    e["colour_p99"] = float(torch.quantile(per_ray, 0.99))
    e["colour_rays_above_1e-4"] = float((per_ray > 1e-4).double().mean())
    if "weights" in got and got["weights"].shape == ref["weights"].shape:  # per-SAMPLE compositing weights [R, S + O]
        e["weights"] = _rel(got["weights"], ref["weights"])
    for k in ("gradients", "cdf_fine"):  # per-SAMPLE SDF gradients (normals) [R, S, 3] and the fine CDF [R, S]: index-wise like `weights`
        if k in got and k in ref and got[k].shape == ref[k].shape:
            e[k] = _rel(got[k], ref[k])
    if "sdf" in got and "sdf" in ref:  # SDF values at the oracle's sample positions: absolute (unit-sphere units) and relative
        e["sdf_abs"] = float((got["sdf"].double().reshape(-1) - ref["sdf"].double().reshape(-1)).abs().max())
        e["sdf"] = _rel(got["sdf"], ref["sdf"])
    return {k: float("%.3g" % v) for k, v in e.items()}


# entry point -> substring of the kernel name the PMC passes report it under
PMC_KERNEL = {"ncw_wgrad_tiled": "wgrad_dma_kernel", "ncw_wgrad": "wgrad_kernel", "ncw_sdf_bwd": "sdf_bwd_kernel",
              "ncw_sdf_fwd": "sdf_fwd", "ncw_nerf_fwd": "nerf_fwd", "ncw_nerf_bwd": "nerf_bwd", "ncw_color_fwd": "color_fwd",
              "ncw_color_bwd": "color_bwd", "ncw_sdf_infer_rays": "sdf_infer"}


def pmc_kernel_name(raw):
    """rocprofv3 kernel name -> short name that keeps anonymous-namespace kernels and template arguments apart."""
    k = raw.replace("(anonymous namespace)::", "").replace("void ", "")
    return k.split("(")[0] if "(" in k else k


def pmc_traffic(argv_inner, steps_inner, timeout=240, env=None, split=False):
    """HBM traffic measured WITH this run: two rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE cannot share a pass:
    MI355X_MICROARCH.md, PMC slots) of a short inner run of this script.  Returns ({kernel: bytes per launch},
    bytes per step over all kernels) or (None, None).  FETCH_SIZE is doubled (the guide's gfx950 rule for wide
    coalesced reads: 16 B/lane global_load and LDS-DMA alike -- every stash / weight read here); both counters are KB."""
    import collections
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    if shutil.which("rocprofv3") is None:
        return None, None
    per_kernel = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(int)
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="ncw_pmc_")
        cmd = ["rocprofv3", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
               os.path.abspath(__file__)] + argv_inner
        try:
            subprocess.run(cmd, cwd=d, env=dict(os.environ if env is None else env, TMPDIR=d), capture_output=True, text=True,
                           timeout=timeout)
            files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
            if not files:
                return None, None
            n = collections.defaultdict(int)
            for row in csv.DictReader(open(files[0])):
                if row["Counter_Name"] != ctr:
                    continue
                k = pmc_kernel_name(row["Kernel_Name"])
                per_kernel[k][ctr] += float(row["Counter_Value"])
                n[k] += 1
            for k, v in n.items():
                launches[k] = max(launches[k], v)
        except Exception:
            return None, None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    out, total, tf, tw = {}, 0.0, 0.0, 0.0
    for k, c in per_kernel.items():
        b = (2.0 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024.0
        total += b
        tf += 2.0 * c.get("FETCH_SIZE", 0.0) * 1024.0
        tw += c.get("WRITE_SIZE", 0.0) * 1024.0
        out[k] = b / max(launches[k], 1)
    if split:  # (fetch bytes, write bytes) per step
        return out, (tf / max(steps_inner, 1), tw / max(steps_inner, 1))
    return out, total / max(steps_inner, 1)


def time_allreduce(numel, dev, world, n_ar=20):
    """In-place all-reduce of a flat fp32 buffer of `numel` elements on the initialised process group, back to back."""
    backend = dist.get_backend()
    op = dist.ReduceOp.AVG if backend == "nccl" else dist.ReduceOp.SUM
    buf = torch.ones(numel, device=dev, dtype=torch.float32)
    for _ in range(3):
        dist.all_reduce(buf, op=op)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    for _ in range(n_ar):
        dist.all_reduce(buf, op=op)
    torch.cuda.synchronize()
    d_ar = (time.perf_counter() - t1) / n_ar
    if world > 1:
        t = torch.tensor([d_ar], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        d_ar = float(t.item())
    return {"allreduce_ms": round(d_ar * 1e3, 4), "bytes": numel * 4, "backend": backend, "world": world,
            "op": "AVG" if backend == "nccl" else "SUM (+ one division launch)",
            "note": "in-place all-reduce of the flat fp32 gradient buffer, back to back, host-timed over %d calls%s"
                    % (n_ar, "; ONE-rank RCCL group: launch + kernel, no wire" if world == 1 else "")}


def allreduce_probe(numel):
    """`bench.py --allreduce-probe N` (child of an N = 1 run): one-rank RCCL group on cuda:0, prints time_allreduce's dict."""
    import socket

    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
    print(json.dumps(time_allreduce(numel, dev, 1)), flush=True)
    dist.destroy_process_group()


def voxel_shell(level=7, r0=0.5, thick=0.05, device="cuda"):
    """the voxels of a level-7 grid that intersect a sphere shell of radius 0.5 +- 0.05."""
    G = 1 << level
    c = (torch.arange(G, device=device).float() + 0.5) * (2.0 / G) - 1.0
    x, y, z = torch.meshgrid(c, c, c, indexing="ij")
    return ((x * x + y * y + z * z).sqrt() - r0).abs() < thick + (2.0 / G)


def bench_grid512(args, nw, L, dev, world, rank):
    """BASELINE configs[4]: the SDF sweep of tools/extract_mesh.py / utils/visualization.py:37-89 -- 512^3 grid points,
    SDF MLP inference only (sdf column), coordinates generated on the chip, contiguous slices per rank."""
    from neuralrecon_w_amd import grid

    W = 512 if args.grid_width is None else args.grid_width
    torch.manual_seed(0)
    net = nw.SDFNetwork(d_in=3, d_out=W + 1, d_hidden=W, n_layers=8, skip_in=(4,), multires=6, bias=0.5, scale=1,
                        geometric_init=True, weight_norm=True, inside_outside=False).to(dev)
    dim = 512
    total = dim ** 3
    start, count, per = grid.local_range(total, rank, world)
    prec = {"bf16": nw.PREC_BF16, "f16": nw.PREC_F16, "f32": nw.PREC_F32}[args.prec]
    out = torch.empty(per, device=dev, dtype=torch.float32)
    lo, hi = (-1.0, -1.0, -1.0), (1.0, 1.0, 1.0)
    chunk = 1 << 22

    def sweep():
        grid.sdf_grid_range(net, dim, lo, hi, start, count, prec=prec, chunk=chunk, out=out)

    for _ in range(max(1, min(args.warmup, 2))):
        sweep()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = max(1, min(args.steps, 3))
    for _ in range(K):
        sweep()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    plain = None
    if prec == nw.PREC_F16 and net.split_value(prec):  # the same sweep without the split-precision value path (4e-4 SDF error)
        net.sdf_split = False
        sweep()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        sweep()
        torch.cuda.synchronize()
        plain = {"ms_per_sweep": (time.perf_counter() - t1) * 1e3, "note": "NEUCONW_SDF_SPLIT=0: one fp16 rounding per operand, "
                 "SDF error 4-6e-4 (moves the zero level set by a tenth of a 512^3 voxel) instead of 5-9e-7"}
        net.sdf_split = None
    # ---- parity of the timed sweep: a random sub-lattice of rank 0's slice against the fp64 oracle (utils/visualization.py:46-50:
    # linspace^3, 'ij' order, x slowest); with --prec f16 this is the split-precision value chain the product defaults to
    parity = None
    if rank == 0 and not args.no_cpu_baseline:
        try:
            from oracle import neuconw_oracle as O

            sweep()
            torch.cuda.synchronize()
            n_s = 32768
            gsel = torch.Generator().manual_seed(7)
            idx = torch.randint(0, count, (n_s,), generator=gsel)
            lin = torch.linspace(lo[0], hi[0], dim, dtype=torch.float64)
            gi = idx + start
            pts = torch.stack([lin[gi // (dim * dim)], lin[(gi // dim) % dim], lin[gi % dim]], -1)
            sd = {"sdf_net." + k: v.detach().cpu().double() for k, v in net.state_dict().items()}
            t_or = time.perf_counter()
            with torch.no_grad():
                ref = O.sdf_net(sd, pts, with_grad=False)[0]
            t_or = time.perf_counter() - t_or
            got = out[idx.to(dev)].cpu().double()
            parity = {"dtype": args.prec, "points": n_s, "of": "rank 0's slice of the 512^3 lattice, random sub-lattice (seed 7)",
                      "oracle": "fp64 torch-CPU oracle (oracle/neuconw_oracle.py sdf_net) at the lattice coordinates of utils/visualization.py:46-50",
                      "sdf_abs": float("%.3g" % float((got - ref).abs().max())),
                      "sdf": float("%.3g" % float((got - ref).abs().max() / ref.abs().max())),
                      "value_path": "split-precision fp16 (hi + lo operands)" if (prec == nw.PREC_F16 and net.split_value(prec)) else args.prec,
                      "oracle_points_per_s_%d_threads" % torch.get_num_threads(): round(n_s / t_or, 1)}
        except Exception as e:
            parity = {"dtype": args.prec, "error": "failed: %r" % (e,)}
    macs = {256: 459008, 512: 1835520}[W]
    pts_s = total * K / dt
    peak = PEAK_BF16_TFLOPS if prec != nw.PREC_F32 else 157.3  # fp16 MFMA peak = bf16 peak
    ach = 2.0 * macs * pts_s / 1e12
    if rank == 0:
        print(json.dumps({
            "metric": "points/sec, 512^3 SDF grid sweep (tools/extract_mesh.py path) [secondary: BASELINE configs[4]]",
            "value": pts_s, "unit": "points/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": args.prec, "data": "synthetic",
            "config": {"workload": "512^3 = 134,217,728 grid points, SDF 8x%d sdf-only inference, coordinates from the linear "
                                   "index on the chip, %d-point launches, contiguous 1/%d slice per rank" % (W, chunk, world),
                       "points_per_rank": per, "sdf_precision_note": "the product default for this path is fp32 "
                       "(NEUCONW_INFER_PREC); this row times --prec; fp16 = the split-precision value path (fp32-level SDF "
                       "values, 3x the MFMAs: `frac` counts algorithmic FLOPs)", "plain_f16": plain},
            "roofline": {"kernel": "ncw_sdf_infer_points", "bound": "mfma", "achieved": round(ach, 1), "peak": peak,
                         "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
                         "algorithmic_mflop_per_point": round(2.0 * macs / 1e6, 3)},
            "parity": parity, "cpu_baseline": None}))


def bench_render(args, nw, L, dev, world, rank):
    """Secondary row `--config render`: the FORWARD-ONLY render of the reference's validation / novel-view path
    (lightning_modules/neuconw_system.py:404-458 -> rendering/renderer.py:785-916 under no_grad): sampler + the three MLPs'
    stash-free kernels + compositor on the headline networks, 1024 rays x (64 + 64) samples per pass (and one
    4096-ray chunk), next to the TRAINING forward of the same batch (grad mode on: full stash)."""
    prec = {"bf16": nw.PREC_BF16, "f16": nw.PREC_F16, "f32": nw.PREC_F32}[args.prec]
    emb, neuconw, nerf, rdr = build_models(dev, prec)
    rdr.bg_dense = not args.bg_eliminate
    R = args.rays
    rays, ts, label, rgbs = synth_batch(R, BATCH_SEED + rank, dev)
    bg = torch.zeros(1, 3, device=dev)
    S = N_SAMPLES + N_IMPORTANCE

    def fwd_only(r=rays, t=ts, l=label):
        with torch.no_grad():
            return rdr.render(r, t, l, background_rgb=bg, cos_anneal_ratio=0.5, perturb_overwrite=0)

    def fwd_train():
        return rdr.render(rays, ts, label, background_rgb=bg, cos_anneal_ratio=0.5, perturb_overwrite=0)

    def timed(fn, k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            o = fn()
            del o
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k

    for _ in range(args.warmup):
        fwd_only()
    if world > 1:
        dist.barrier()
    dt = timed(fwd_only, args.steps)
    if world > 1:
        dist.barrier()
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if args.inner:
        if rank == 0:
            print(json.dumps({"inner": True, "ms_per_step": dt * 1e3}))
        return
    for _ in range(3):
        o = fwd_train()
        del o
    dt_train = timed(fwd_train, args.steps)
    same = None
    a_, b_ = fwd_train(), fwd_only()
    same = all(torch.equal(a_[k].detach(), b_[k]) for k in ("color", "depth", "weights", "weights_sum", "gradients", "cdf_fine", "color_bg"))
    del a_, b_
    # per-kernel HIP-event times of both forms (one stream)
    two, rdr.use_bg_stream = rdr.use_bg_stream, False
    per = {}
    for name, fn in (("forward_only", fwd_only), ("training_forward", fwd_train)):
        L.PROFILE = {}
        for _ in range(3):
            o = fn()
            del o
        torch.cuda.synchronize()
        prof, L.PROFILE = L.PROFILE, None
        prof = {(k[:-4] if k.endswith("_f16") else k): v for k, v in prof.items()}
        per[name] = {k: round(sum(a.elapsed_time(b) for a, b in v) / 3, 4) for k, v in sorted(prof.items())}
    rdr.use_bg_stream = two
    arena = {}
    for nm, mod in (("sdf", neuconw.sdf_net), ("colour", neuconw.color_net), ("background", nerf)):
        for key, e in mod.__dict__["_stash_cache"]._e.items():
            arena["%s_%s_mb" % (nm, "training" if key[-1] else "forward_only")] = round(e["arena"].buf.numel() / 1e6, 2)
    # one 4096-ray chunk (the validation loop renders an image in chunks)
    r4, t4, l4, _ = synth_batch(4096, 77, dev)
    for _ in range(2):
        fwd_only(r4, t4, l4)
    dt4 = timed(lambda: fwd_only(r4, t4, l4), max(2, args.steps // 4))
    traffic = None
    if not args.no_pmc and rank == 0:
        inner = ["--inner", "--config", "render", "--gpus", "1", "--steps", "3", "--warmup", "2", "--prec", args.prec, "--rays", str(R),
                 "--no-cpu-baseline", "--no-pmc"] + (["--bg-eliminate"] if args.bg_eliminate else [])
        per_kernel, step_bytes = pmc_traffic(inner, 5, timeout=240, split=True)
        if per_kernel:
            traffic = {"fetch_gb_per_render": round(step_bytes[0] / 1e9, 3), "write_gb_per_render": round(step_bytes[1] / 1e9, 3),
                       "note": "rocprofv3 --pmc passes of this row (FETCH_SIZE x 2 per the gfx950 rule; WRITE_SIZE as counted); the writes "
                               "are the SDF network's h_l scratch (re-read by its own adjoint sweep: 8 x 512 B per sample = 0.54 GB) + feat",
                       "kernel_mb_per_launch": {k[:48]: round(v / 1e6, 1) for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1])[:8]}}
    parity = None
    if rank == 0 and not args.no_cpu_baseline:
        try:
            ref = oracle_outputs()
            parity = {"dtype": args.prec, "rays": 256, "oracle": "fp64 oracle, first 256 rays of the batch, inv_s 20",
                      "bitwise_equal_to_training_forward": bool(same)}
            parity.update(parity_errors(gpu_outputs(dev, prec, pts=ref["pts"]), ref))
        except Exception as e:
            parity = {"error": "failed: %r" % (e,), "bitwise_equal_to_training_forward": bool(same)}
    if rank == 0:
        mlp = ("ncw_sdf_fwd", "ncw_color_fwd", "ncw_nerf_fwd")
        print(json.dumps({
            "metric": "ray-samples/sec (forward-only render) at %d rays x %d samples [secondary: validation / novel-view path]" % (R, S),
            "value": world * R * S / dt, "unit": "ray-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.prec,
            "data": "synthetic",
            "config": {"workload": "forward-only render (torch.no_grad): sampler + SDF / colour / background MLPs (stash-free kernels) + "
                                   "compositor, %d rays x (%d + %d) samples, headline networks, background %s"
                                   % (R, N_SAMPLES, N_IMPORTANCE, "on every sample" if rdr.bg_dense else "eliminated where dead"),
                       "rays_per_gpu": R, "samples_per_ray": S},
            "training_forward_ms": dt_train * 1e3, "forward_only_over_training_forward": round(dt / dt_train, 4),
            "mlp_forward_ms": {k: round(sum(per[k].get(m, 0.0) for m in mlp), 4) for k in per},
            "per_kernel_ms": per, "stash_arena": arena, "chunk_4096_rays_ms": dt4 * 1e3,
            "chunk_4096_rays_value": 4096 * S / dt4, "traffic": traffic, "parity": parity, "roofline": None, "cpu_baseline": None}))


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--allreduce-probe":
        return allreduce_probe(int(sys.argv[2]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--prec", default="f16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--rays", type=int, default=None,
                    help="rays per GPU (default 1024 = the metric's batch; --config shipped: 2048 = scripts/train.sh:16-19)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc passes behind roofline.traffic")
    ap.add_argument("--bg-eliminate", action="store_true",
                    help="secondary: time the MAIN leg with dead-background elimination (the product default) instead of "
                         "evaluating the background NeRF on every sample like the reference; marked in the metric name")
    ap.add_argument("--no-parity-mode", action="store_true", help="skip the fp32 parity-mode timing (`parity_mode`)")
    ap.add_argument("--seed", type=int, default=1000, help="seed of the synthetic ray batch (default 1000: the reported line); the parity legs follow it")
    ap.add_argument("--save-trained-state", default=None, help="write the trained-weights parity point's state_dict here (.pt)")
    ap.add_argument("--inner", action="store_true", help="(internal) the short run the PMC passes profile: timing loop only")
    ap.add_argument("--grid-width", type=int, default=None, choices=[256, 512], help="--config grid512: SDF width (default 512)")
    ap.add_argument("--config", default="headline", choices=["headline", "shipped", "voxel", "grid512", "render"],
                    help="headline = BASELINE.json configs[1] (the metric's shape).  Secondary rows, never the reported "
                         "metric: shipped = the reference's yaml shape (W=512 SDF, 8+16 samples); voxel = configs[2] "
                         "(headline + level-7 shell occupancy: voxel near/far, +-16-voxel window, 10 boundary samples); "
                         "grid512 = configs[4] (512^3 SDF sweep, points/s)")
    ap.add_argument("--graph", action="store_true",
                    help="record the step into HIP graphs and replay it (trainer.TrainStep(capture=True)); measured "
                         "4.78 vs 4.80 ms eager on one MI355X -- the step is not host-launch-bound -- so eager is the default")
    ap.add_argument("--one-stream", action="store_true",
                    help="run the background NeRF's launches on the main stream too (NEUCONW_BG_STREAM=0): nothing overlaps, so a rocprofv3 "
                         "kernel trace of this run reproduces `per_step_kernel_ms` kernel by kernel (profiles/r06/bench_onestream_*)")
    ap.add_argument("--dist-check", action="store_true",
                    help="initialise the process group, print {world, ranks, devices} from rank 0 and exit (launcher test; "
                         "needs no GPU with NCW_DIST_BACKEND=gloo)")
    args = ap.parse_args()
    if args.one_stream:  # read when the renderer is constructed (and inherited by the PMC / parity child processes)
        os.environ["NEUCONW_BG_STREAM"] = "0"
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
    globals().update(BATCH_SEED=int(args.seed))
    if args.config == "shipped":  # config/train_brandenburg_gate.yaml: SDF 8x512, N_SAMPLES 8, N_IMPORTANCE 16
        globals().update(W_SDF=512, N_SAMPLES=8, N_IMPORTANCE=16, M_SDF=2097664, M_SDF1=1835520, M_COL=585344)
    if args.rays is None:  # the reference's recipe trains 2048 rays per GPU (scripts/train.sh:16-19)
        args.rays = 2048 if args.config == "shipped" else R_PER_GPU

    import neuralrecon_w_amd as nw
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
        import datetime

        # rank 0 runs the CPU baseline / parity / PMC legs alone while the other ranks wait in a barrier (1-4 min): give the
        # collective watchdog room
        tmo = datetime.timedelta(minutes=30)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)
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
    if args.config == "grid512":
        bench_grid512(args, nw, L, dev, world, rank)
        if world > 1:
            dist.destroy_process_group()
        return
    if args.config == "render":
        bench_render(args, nw, L, dev, world, rank)
        if world > 1:
            dist.destroy_process_group()
        return
    prec = {"bf16": nw.PREC_BF16, "f16": nw.PREC_F16, "f32": nw.PREC_F32}[args.prec]
    R = args.rays
    bg = torch.zeros(1, 3, device=dev)
    rays, ts, label, rgbs = synth_batch(R, BATCH_SEED + rank, dev)
    n_boundary = 0

    def make_step(prec_, bg_dense=True, sdf_split=None):
        """models + TrainStep in precision prec_ -> step(i).  bg_dense=True: the background NeRF on every one of the
        S + O samples of a ray like the reference evaluates it (the timed `value`); False = the product default, which
        skips the samples whose result the compositor multiplies by 0 (reported as `bg_elimination`).  LR rule of train.py:21-25: 1e-4 * world*batch / 4096; Adam
        eps 1e-7 (utils/__init__.py:24-31); clip 0.99 (train.py:61).  TrainStep = render + loss + backward + one flat
        all-reduce + clip + Adam (trainer.py)."""
        emb_, neuconw_, nerf_, rdr_ = build_models(dev, prec_)
        rdr_.bg_dense = bg_dense
        if sdf_split is not None:  # None = the product default (fp16: split-precision SDF value path, csrc/ncw_split.hip)
            neuconw_.sdf_net.sdf_split = sdf_split
            if not sdf_split:  # `plain_f16_mode`: no split-precision work anywhere a switch exists (also the background refinement)
                nerf_.refine = False
        if args.config == "voxel":  # configs[2]: coarse octree -> ray near/far; fine octree -> +-SAMPLE_RANGE window + boundary samples
            voxel_setup(rdr_, dev)
        train_ = nw.TrainStep(rdr_, [emb_, neuconw_, nerf_], loss_fn, lr=1e-4 * world * R / 4096.0, eps=1e-7, clip=0.99,
                              world_size=world, capture=args.graph, capture_warmup=3)

        def step_(i):
            loss_, _ = train_(rays, ts, label, rgbs, background_rgb=bg, cos_anneal_ratio=min(1.0, i / 50000.0))
            return loss_

        return step_, train_, (emb_, neuconw_, nerf_, rdr_)

    if args.config == "voxel":
        n_boundary = 10
        globals().update(N_BOUNDARY=10)
    step, train, (emb, neuconw, nerf, rdr) = make_step(prec, bg_dense=not args.bg_eliminate)

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
    S = N_SAMPLES + N_IMPORTANCE + n_boundary
    value = world * R * S * args.steps / dt
    # a step whose gradient norm is not finite is SKIPPED by ncw_adam_step_dev (fp16 loss-scale guard) and is cheaper than a
    # real one: the line reports the count over warm-up + timed steps and the bench FAILS if any step was skipped
    skipped = int(train.opt.skipped_steps) if hasattr(train.opt, "skipped_steps") else int(getattr(train, "skipped_steps", 0))
    if world > 1:
        t = torch.tensor([skipped], device=dev, dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        skipped = int(t.item())
    if args.inner:  # the PMC passes only need the kernels to run
        if rank == 0:
            print(json.dumps({"inner": True, "ms_per_step": dt / args.steps * 1e3}))
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- per-kernel HIP-event timing over a few more live steps (same stream) -> roofline ---------------
    roofline = None
    # every rank runs the same extra steps (the step contains the gradient all-reduce); only rank 0 records
    prof_steps = max(2, min(5, args.steps))
    if rank == 0:
        L.PROFILE = {}
    # The timed region above runs the background NeRF's launches on a second stream beside the SDF / colour chain
    # (renderer.use_bg_stream); an event pair around a launch that shares the device measures both chains.  The per-kernel
    # pass therefore runs the same steps on ONE stream: every duration below is the kernel's own (the dominant kernel, the
    # weight-gradient launch, runs alone in either form, so its duration is also what rocprofv3 reports for the timed region).
    two_streams, rdr.use_bg_stream = rdr.use_bg_stream, False
    for i in range(prof_steps):
        train.eager_step(rays, ts, label, rgbs, background_rgb=bg,
                         cos_anneal_ratio=min(1.0, (args.warmup + args.steps + i) / 50000.0))
    torch.cuda.synchronize()
    rdr.use_bg_stream = two_streams
    if world > 1:
        dist.barrier()
    if rank == 0:
        prof, L.PROFILE = L.PROFILE, None
        prof = {(k[:-4] if k.endswith("_f16") else k): v for k, v in prof.items()}  # the fp16 build of an entry point
        fl = kernel_flops(R)
        rows = {}
        for name, evs in prof.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            rows[name] = (sum(ms) / prof_steps, len(evs) // prof_steps, ms)
        total_ms = sum(v[0] for v in rows.values())
        dom = max((k for k in rows if fl.get(k) is not None), key=lambda k: rows[k][0])
        avg_ms = rows[dom][0] / rows[dom][1]
        ach = fl[dom] / (rows[dom][0] * 1e-3) / 1e12
        peak = PEAK_BF16_TFLOPS if prec != nw.PREC_F32 else 157.3  # fp16 MFMA peak = bf16 peak
        frac_mfma = ach / peak
        roofline = {"kernel": dom, "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(frac_mfma, 4), "frac_mfma": round(frac_mfma, 4), "traffic": None,
                    "avg_launch_ms": round(avg_ms, 4), "launches_per_step": rows[dom][1],
                    "algorithmic_gflop_per_launch": round(fl[dom] / rows[dom][1] / 1e9, 2),
                    "kernel_tflops": {k: round(fl[k] / (rows[k][0] * 1e-3) / 1e12, 1) for k in rows if fl.get(k)},
                    "kernel_frac_mfma": {k: round(fl[k] / (rows[k][0] * 1e-3) / 1e12 / peak, 4) for k in rows if fl.get(k)},
                    "per_step_kernel_ms": {k: round(v[0], 4) for k, v in sorted(rows.items(), key=lambda kv: -kv[1][0])},
                    "sum_kernel_ms_per_step": round(total_ms, 4),
                    "kernel_timing": "HIP events per launch over %d more live steps on one stream (the timed region overlaps "
                                     "the background NeRF's launches with the SDF / colour chain on a second stream: "
                                     "ms_per_step < sum_kernel_ms_per_step)" % prof_steps}
        if dom.startswith("ncw_wgrad"):
            # The weight-gradient GEMMs reduce over the POINTS: every product streams its two stash operands
            # once (algorithmic bytes = sum over products of (rbx + rby) x 32 features x elem x points); at
            # ~100 FLOP/B they sit left of the ridge (2.5 PFLOP/s / 8 TB/s = 312 FLOP/B): HBM-bound in THIS design.
            # SURVEY 8(d) prices the MLP backward against MFMA: both fractions are reported, `frac` follows `bound`.
            esz = 2 if prec != nw.PREC_F32 else 4
            alg_bytes = 0.0
            for ent in neuconw.sdf_net.__dict__.get("_stash_cache")._e.values():
                wb = ent.get("wgrad_batch")
                if wb is not None:
                    alg_bytes += type(wb).algorithmic_bytes(wb.items, esz)
            gbs = alg_bytes / (rows[dom][0] * 1e-3) / 1e9
            # `frac` stays the SURVEY 8(d) fraction (algorithmic FLOPs / duration / MFMA peak: 8d classes the MLP backward
            # as MFMA-bound); the bytes THIS DESIGN streams through the kernel are reported beside it as `frac_hbm_design`
            roofline.update({"bound": "mfma", "bound_note": "SURVEY 8(d) prices the MLP backward against the MFMA peak; this "
                             "design's weight-gradient launch streams its stash operands once and is HBM-bound at `frac_hbm_design`",
                             "frac_hbm_design": round(gbs / 8000.0, 4), "design_gbytes_per_s": round(gbs, 1), "hbm_peak_gbytes_per_s": 8000.0,
                             "design_gbytes_per_launch": round(alg_bytes / 1e9, 3), "mfma_tflops": round(ach, 1)})
        # step-level MFMA fraction: all algorithmic FLOPs of the step / wall time
        step_flops = 2.0 * R * ((N_SAMPLES + (UP_STEPS - 1) * N_IMPORTANCE // UP_STEPS) * M_SDF1
                                + S * (6 * M_SDF + 3 * M_COL) + (S + N_OUTSIDE) * 3 * M_BG)
        roofline["step_algorithmic_tflop"] = round(step_flops / 1e12, 4)
        # SURVEY 8(d)'s algorithmic HBM bytes: 8.3 KB per ray-sample (inputs + outputs + ONE 8 x W x 2 B activation stash)
        roofline["survey_algorithmic_gbytes_per_step"] = round(8.3e3 * R * S / 1e9, 3)
        roofline["step_frac_of_mfma_peak"] = round(step_flops / (dt / args.steps) / 1e12 / peak, 4)
        # ---- HBM traffic, measured with this run (single-GPU runs; N > 1 would profile N ranks) ----------------
        if not args.no_pmc:
            # N > 1: the passes profile ONE single-process replica of the per-rank step on rank 0's device (the other ranks
            # wait at the barrier below): the per-GPU kernels are the same at every N, the all-reduce is not in `traffic`
            inner_steps, inner_warm = 3, 2
            inner = ["--inner", "--gpus", "1", "--steps", str(inner_steps), "--warmup", str(inner_warm), "--prec", args.prec,
                     "--rays", str(R), "--config", args.config, "--no-cpu-baseline", "--no-pmc", "--no-parity-mode"]
            if args.bg_eliminate:
                inner.append("--bg-eliminate")
            env1 = None
            if world > 1:
                env1 = {k: v for k, v in os.environ.items()
                        if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR",
                                     "MASTER_PORT", "TORCHELASTIC_RUN_ID", "GROUP_WORLD_SIZE", "ROLE_WORLD_SIZE", "ROLE_NAME")}
                vis = os.environ.get("HIP_VISIBLE_DEVICES")
                env1["HIP_VISIBLE_DEVICES"] = vis.split(",")[local_rank] if vis else str(local_rank)
            per_kernel, step_bytes = pmc_traffic(inner, inner_steps + inner_warm, timeout=240 if world == 1 else 150, env=env1)
            if per_kernel:
                sel = [v for k, v in per_kernel.items() if PMC_KERNEL.get(dom, "\0") in k]
                roofline["traffic"] = round(max(sel), 0) if sel else None
                roofline["traffic_source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this run (FETCH x 2 + WRITE)"
                roofline["step_traffic_gb"] = round(step_bytes / 1e9, 3)
                roofline["traffic_ratio"] = round(step_bytes / (8.3e3 * R * S), 2)  # measured step bytes / SURVEY 8(d) bytes
                top = sorted(((k, v) for k, v in per_kernel.items()), key=lambda kv: -kv[1])[:10]
                roofline["kernel_traffic_mb_per_launch"] = {k[:48]: round(v / 1e6, 1) for k, v in top}

    # ---- the gradient all-reduce on its own: the flat fp32 gradient buffer, in place, ReduceOp.AVG on RCCL (what
    # FlatParams.allreduce issues once per step).  At N = 1 a one-rank RCCL group is created just for this: it prices the
    # launch + the in-place kernel, not the wire; at N > 1 it is the real collective over xGMI (max over ranks).
    allreduce = None
    try:
        if world == 1 and not dist.is_initialized():
            # a ONE-rank RCCL group, in a child process with a hard time limit: a backend that cannot come up on this box
            # must cost the bench line a null, never a hang
            import subprocess

            r_ = subprocess.run([sys.executable, os.path.abspath(__file__), "--allreduce-probe", str(train.fp.flat_grad.numel())],
                                capture_output=True, text=True, timeout=90, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
            ls_ = [l for l in r_.stdout.splitlines() if l.startswith("{")]
            allreduce = json.loads(ls_[-1]) if ls_ else {"allreduce_ms": None, "error": (r_.stderr or r_.stdout)[-300:]}
        else:
            allreduce = time_allreduce(train.fp.flat_grad.numel(), dev, world)
    except Exception as e:  # never take the bench line down
        allreduce = {"allreduce_ms": None, "error": "%r" % (e,)}

    # ---- the fp32 parity mode (the <= 1e-4 mode, tests/test_gpu_render.py) timed in the same process ---------------
    parity = None
    alt = elim = plain = None
    if not args.no_parity_mode and args.prec in ("bf16", "f16") and not args.graph:
        del train, step
        torch.cuda.empty_cache()
        alt_name = "bf16" if args.prec == "f16" else "f16"
        step_a, train_a, _ = make_step({"bf16": nw.PREC_BF16, "f16": nw.PREC_F16}[alt_name])
        for i in range(5):
            step_a(i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            step_a(5 + i)
        torch.cuda.synchronize()
        d_a = (time.perf_counter() - t1) / args.steps
        alt = {"dtype": alt_name, "value": R * S / d_a, "unit": "ray-samples/s", "ms_per_step": d_a * 1e3, "steps": args.steps}
        del train_a, step_a
        torch.cuda.empty_cache()
        if args.prec == "f16":  # what the split-precision SDF value path costs: the same step without it
            step_p, train_p, _ = make_step(prec, sdf_split=False)
            for i in range(5):
                step_p(i)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(args.steps):
                step_p(5 + i)
            torch.cuda.synchronize()
            d_p = (time.perf_counter() - t1) / args.steps
            plain = {"dtype": "f16", "value": R * S / d_p, "unit": "ray-samples/s", "ms_per_step": d_p * 1e3, "steps": args.steps,
                     "note": "NEUCONW_SDF_SPLIT=0 NEUCONW_NERF_REFINE=0: one fp16 rounding per operand in the SDF value chain and "
                             "in the background NeRF (round 2's kernels); rendered outputs 5e-4 of the oracle at inv_s 20 and 2e-2 at inv_s 403 on these rays, "
                             "against 1e-5 / 5e-5 for the timed path"}
            del train_p, step_p
            torch.cuda.empty_cache()
        # the product default: dead-background elimination (renderer.py _RenderFn.forward) -- identical outputs and
        # gradients (tests/test_gpu_bg_select.py), the NeRF evaluated only where the compositor can use it
        step_e, train_e, (_, _, _, rdr_e) = make_step(prec, bg_dense=False)
        for i in range(5):
            step_e(i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            step_e(5 + i)
        torch.cuda.synchronize()
        d_e = (time.perf_counter() - t1) / args.steps
        with torch.no_grad():
            o_e = rdr_e.render(rays, ts, label, background_rgb=bg, cos_anneal_ratio=0.0)
        elim = {"value": R * S / d_e, "unit": "ray-samples/s", "ms_per_step": d_e * 1e3, "steps": args.steps, "dtype": args.prec,
                "inside_fraction": float(o_e["inside_sphere"].float().mean()),
                "note": "background NeRF evaluated only on primary samples outside the unit sphere + the outside samples "
                        "(the rest is multiplied by 1 - inside_sphere = 0 in the compositor, renderer.py:693-708); outputs "
                        "bitwise identical, gradients to summation order; `value` above evaluates all S + O samples like "
                        "the reference"}
        del train_e, step_e, rdr_e, o_e
        torch.cuda.empty_cache()
        step32, train32, _ = make_step(nw.PREC_F32)
        for i in range(2):
            step32(i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        k32 = 5
        for i in range(k32):
            step32(2 + i)
        torch.cuda.synchronize()
        d32 = (time.perf_counter() - t1) / k32
        del train32, step32
        torch.cuda.empty_cache()
        step32e, train32e, _ = make_step(nw.PREC_F32, bg_dense=False)  # the fp32 mode as the product runs it
        for i in range(2):
            step32e(i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(k32):
            step32e(2 + i)
        torch.cuda.synchronize()
        d32e = (time.perf_counter() - t1) / k32
        parity = {"dtype": "f32", "value": R * S / d32, "unit": "ray-samples/s", "ms_per_step": d32 * 1e3, "steps": k32,
                  "bg_elimination_ms_per_step": d32e * 1e3,
                  "note": "exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, 157 TFLOP/s peak): outputs within 1e-4 of the reference, "
                          "bitwise run-to-run reproducible; step_frac_of_f32_mfma_peak = %.3f"
                          % (roofline["step_algorithmic_tflop"] / d32 / 157.3 if roofline else float("nan"))}

    cpu = None
    parity_obj = None
    if rank == 0 and not args.no_cpu_baseline:  # N > 1: rank 0 alone (the other ranks wait at the barrier below)
        try:
            cpu, ref32 = cpu_baseline()
        except Exception as e:  # the baseline must never take the bench line down
            cpu, ref32 = {"value": None, "unit": "ray-samples/s", "cores": os.cpu_count(), "kind": "port",
                          "sample": "failed: %r" % (e,)}, None
        # ---- measured error of the program that was timed: the GPU render + loss of the oracle leg's 256 rays (same
        # batch, same initial weights, deterministic sampling) in the TIMED dtype, against the oracle -- at the initial
        # operating point (variance 0.3, inv_s 20) and where NeuS trains (variance 0.6: inv_s = exp(6) = 403)
        if args.config == "voxel":  # configs[2] in the TIMED dtype against the fp64 oracle with the same coarse + fine octree
            try:
                ref_v = oracle_outputs(voxel=True)
                parity_obj = {"dtype": args.prec, "rays": 256, "measure": "max|gpu - oracle| / max|oracle| (loss: absolute)",
                              "oracle": "fp64 oracle with the same level-7 shell occupancy as coarse (ray near / far) and fine (+-16-voxel "
                                        "window, 10 boundary samples) octree; kaolin's ray / voxel query itself is UNPINNED (restated)"}
                parity_obj.update(parity_errors(gpu_outputs(dev, prec, pts=ref_v["pts"], voxel=True), ref_v))
                ref_vt = oracle_outputs(variance=0.6, voxel=True)
                parity_obj["at_inv_s_403"] = parity_errors(gpu_outputs(dev, prec, variance=0.6, pts=ref_vt["pts"], voxel=True), ref_vt)
                if prec != nw.PREC_F32:
                    parity_obj["f32_mode"] = parity_errors(gpu_outputs(dev, nw.PREC_F32, pts=ref_v["pts"], voxel=True), ref_v)
            except Exception as e:
                parity_obj = {"dtype": args.prec, "error": "failed: %r" % (e,)}
        if ref32 is not None and args.config in ("headline", "shipped"):
            try:
                parity_obj = {"dtype": args.prec, "rays": 256, "measure": "max|gpu - oracle| / max|oracle| (loss: absolute)",
                              "oracle": "fp32 torch-CPU oracle of the cpu_baseline leg (inv_s 20); fp64 oracle at inv_s 403"}
                # what the UNMODIFIED reference's own fp32 arithmetic differs from the fp64 oracle by on these same 256 rays
                # (build container, scripts/diag/port_over_reference.py family -> profiles/r04/port_over_reference.json)
                parity_obj["reference_fp32_vs_fp64_oracle_same_rays"] = recorded_calibration()[1]
                parity_obj["outputs"] = ("colour / depth / weights_sum per ray; `weights` = per-SAMPLE compositing weights [R, S+O]; "
                                         "`sdf` = SDF network at the oracle's sample positions (sdf_abs in unit-sphere units)")
                parity_obj.update(parity_errors(gpu_outputs(dev, prec, pts=ref32["pts"]), ref32))
                # the comparison that isolates the MLPs + compositor from the discrete sampler: the GPU evaluates the ORACLE's own
                # primary sample depths (render(_z_override=...)); per-SAMPLE `weights` are index-aligned here by construction
                fz = parity_errors(gpu_outputs(dev, prec, z_override=ref32["z_vals"]), ref32)
                parity_obj["fixed_z"] = {k: fz[k] for k in ("colour", "depth", "weights_sum", "weights", "gradients", "cdf_fine") if k in fz}
                parity_obj["fixed_z"]["note"] = ("GPU MLPs + compositor at the oracle's z_vals (no sampler in the comparison): the "
                                                 "per-sample `weights` figure the north star's 1e-4 applies to")
                ref_t = oracle_outputs(variance=0.6)
                parity_obj["at_inv_s_403"] = parity_errors(gpu_outputs(dev, prec, variance=0.6, pts=ref_t["pts"]), ref_t)
                fz = parity_errors(gpu_outputs(dev, prec, variance=0.6, z_override=ref_t["z_vals"]), ref_t)
                parity_obj["at_inv_s_403"]["fixed_z"] = {k: fz[k] for k in ("colour", "depth", "weights_sum", "weights") if k in fz}
                # trained weights: 40 fp32 TrainSteps on these rays, then variance 0.6 (tests/test_gpu_fullsize.py)
                st_tr = trained_state(dev)
                if args.save_trained_state:  # for scripts/diag/reference_on_trained_state.py (the reference's own fp32 on these weights)
                    os.makedirs(os.path.dirname(os.path.abspath(args.save_trained_state)), exist_ok=True)
                    torch.save(st_tr, args.save_trained_state)
                ref_tr = oracle_outputs(state=st_tr)
                parity_obj["trained_40_steps_inv_s_403"] = parity_errors(gpu_outputs(dev, prec, pts=ref_tr["pts"], state=st_tr), ref_tr)
                fz = parity_errors(gpu_outputs(dev, prec, state=st_tr, z_override=ref_tr["z_vals"]), ref_tr)
                parity_obj["trained_40_steps_inv_s_403"]["fixed_z"] = {k: fz[k] for k in ("colour", "depth", "weights_sum", "weights",
                                                                                          "colour_rays_above_1e-4") if k in fz}
                if prec != nw.PREC_F32:
                    parity_obj["f32_mode"] = parity_errors(gpu_outputs(dev, nw.PREC_F32, pts=ref32["pts"]), ref32)
                    parity_obj["f32_mode_at_inv_s_403"] = parity_errors(gpu_outputs(dev, nw.PREC_F32, variance=0.6, pts=ref_t["pts"]), ref_t)
                    parity_obj["f32_mode_trained_40_steps_inv_s_403"] = parity_errors(
                        gpu_outputs(dev, nw.PREC_F32, pts=ref_tr["pts"], state=st_tr), ref_tr)
            except Exception as e:
                parity_obj = {"dtype": args.prec, "error": "failed: %r" % (e,)}

    if world > 1:
        dist.barrier()  # rank 0's CPU-baseline / parity / PMC legs are over
    if rank == 0:
        names = {"headline": "BASELINE.json configs[1]", "shipped": "shipped yaml shape, secondary",
                 "voxel": "BASELINE.json configs[2] (voxel-guided), secondary"}
        line = {
            "metric": ("ray-samples/sec (train step) at 1024 rays x 128 samples [secondary: dead-background elimination]"
                       if args.bg_eliminate and args.config == "headline" else
                       "ray-samples/sec (train step) at 1024 rays x 128 samples" if args.config == "headline" else
                       "ray-samples/sec (train step) at %d rays x %d samples [secondary shape]" % (R, S)), "value": value,
            "unit": "ray-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.prec, "data": "synthetic",
            "config": {"workload": "brandenburg_gate config (%s): %d rays/GPU x (%d coarse + %d fine%s) "
                                   "samples, SDF 8x%d + colour 4x256 + bg NeRF 8x256, 4 outside samples, up_sample_steps 2, "
                                   "render+loss+backward+allreduce+clip+Adam"
                                   % (names[args.config], R, N_SAMPLES, N_IMPORTANCE,
                                      " + 10 boundary, level-7 shell occupancy" if args.config == "voxel" else "", W_SDF),
                       "rays_per_gpu": R, "samples_per_ray": S, "global_rays": world * R, "parallelism": "dp%d" % world,
                       "world_size": world, "ranks": ranks,
                       "submission": "hip-graph replay" if args.graph else "eager",
                       "streams": 1 if os.environ.get("NEUCONW_BG_STREAM", "1") in ("0", "") else 2,
                       "final_loss": float(loss.detach()), "skipped_steps": skipped},
            "allreduce": allreduce, "allreduce_ms": allreduce.get("allreduce_ms") if allreduce else None,
            "roofline": roofline, "parity": parity_obj, "parity_mode": parity, "alt_mode": alt, "plain_f16_mode": plain,
            "bg_elimination": elim,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()
    if skipped > 0:
        raise SystemExit("bench.py: %d of the %d warm-up + timed steps were SKIPPED by the non-finite-gradient guard (fp16 loss "
                         "scale): the timed region is not %d real steps" % (skipped, args.warmup + args.steps, args.steps))


if __name__ == "__main__":
    main()
