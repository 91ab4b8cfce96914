"""Build-container only (needs /root/reference): two calibrations that bench.py / the tests cite but cannot run on the GPU box.

(1) `port_over_reference`: bench.py's `cpu_baseline` times the ORACLE (kind "port": the analytic-adjoint restatement, 6 M_sdf
    per sample) because the GPU box has no reference tree.  Here the UNMODIFIED reference (rendering/renderer.py render +
    losses.py NeuconWLoss + backward through its double forward + autograd.grad, ~9 M_sdf) and the oracle are timed on the
    SAME 256 rays of the bench batch, same networks, same thread count; the ratio goes into bench.py (PORT_OVER_REFERENCE).
(2) `fp32 noise floor`: the reference ITSELF in fp32 against the fp64 oracle on the same rays at variance 0.3 / 0.6 / 0.7
    (inv_s 20 / 403 / 1097), initial weights and weight_v-jittered weights: how far two fp32-accurate evaluations of the same
    function are apart where NeuS trains.  tests/test_gpu_fullsize.py takes TRAINED_TOL from this file, not from our kernels.

    python scripts/diag/port_over_reference.py [--threads 8] -> profiles/r04/port_over_reference.json
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bench  # noqa: E402
from oracle import neuconw_oracle as O  # noqa: E402
from oracle import ref_import  # noqa: E402
import make_golden as G  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--threads", type=int, default=8)
ap.add_argument("--rays", type=int, default=256)
ap.add_argument("--repeats", type=int, default=3)
ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04", "port_over_reference.json"))
args = ap.parse_args()
torch.set_num_threads(args.threads)
ns = ref_import.load()


def reference_system(sd):
    """The reference's own modules (W = 256 headline networks) carrying the state_dict `sd`."""
    emb, neuconw, nerf, rdr = G.build_reference(ns, bench.W_SDF, 8, (4,), n_a=bench.N_A, n_vocab=sd["embedding_a.weight"].shape[0], nerf_w=256,
                                                color_hidden=256, head=128, seed=0, n_samples=bench.N_SAMPLES,
                                                n_importance=bench.N_IMPORTANCE)
    with torch.no_grad():
        emb.weight.copy_(sd["embedding_a.weight"])
        neuconw.load_state_dict({k[len("neuconw."):]: v for k, v in sd.items() if k.startswith("neuconw.")}, strict=False)
        nerf.load_state_dict({k[len("nerf."):]: v for k, v in sd.items() if k.startswith("nerf.")})
    return emb, neuconw, nerf, rdr


cfg_loss = G.AttrDict(NEUCONW=G.AttrDict(MESH_MASK_LIST=["sky"], DEPTH_LOSS=True, FLOOR_NORMAL=False))
ref_loss = ns.NeuconWLoss(coef=1.0, igr_weight=1e-4, mask_weight=0.1, depth_weight=0.1, floor_weight=0.0, config=cfg_loss)
res = {"threads": args.threads, "rays": args.rays, "host": os.uname().nodename, "cpu_count": os.cpu_count()}

# ---- (1) timing: the unmodified reference beside the oracle ------------------------------------------------------
sd0, cfg, (rays, ts, label, rgbs) = bench._oracle_setup(args.rays, 1000)
emb, neuconw, nerf, rdr = reference_system(sd0)
params = [p for m in (emb, neuconw, nerf) for p in m.parameters()]
t_ref = []
for i in range(args.repeats + 1):
    t0 = time.perf_counter()
    out = rdr.render(rays.clone(), ts, label, perturb_overwrite=0, background_rgb=torch.zeros(1, 3), cos_anneal_ratio=0.5)
    loss = sum(ref_loss(out, rgbs).values())
    torch.autograd.grad(loss, params, allow_unused=True)
    t_ref.append(time.perf_counter() - t0)
sdg = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd0.items()}
t_or = []
for i in range(args.repeats + 1):
    t0 = time.perf_counter()
    o = O.render(sdg, cfg, rays, ts, label, 0.5, torch.zeros(1, 3))
    lo = O.neuconw_loss(o, rgbs, cfg)
    torch.autograd.grad(lo, [v for v in sdg.values() if v.requires_grad], allow_unused=True)
    t_or.append(time.perf_counter() - t0)
med = lambda v: sorted(v[1:])[len(v[1:]) // 2]  # noqa: E731
S = bench.N_SAMPLES + bench.N_IMPORTANCE
res["timing"] = {"reference_s_per_step": med(t_ref), "oracle_s_per_step": med(t_or),
                 "reference_ray_samples_per_s": args.rays * S / med(t_ref), "oracle_ray_samples_per_s": args.rays * S / med(t_or),
                 "port_over_reference": med(t_ref) / med(t_or), "loss_reference": float(loss), "loss_oracle": float(lo),
                 "note": "same %d rays of bench.py's batch, same initial weights, perturb 0, render + NeuconWLoss + backward, "
                         "median of %d after 1 warm-up, %d threads" % (args.rays, args.repeats, args.threads)}
print(json.dumps(res["timing"]))


# ---- (2) noise floor: the reference in fp32 vs the fp64 oracle at trained sharpness ------------------------------
def rel(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


floor = []
R2 = 64
for v_jit in (0.0, 0.05):
    sd = {k: v.clone() for k, v in sd0.items()}
    if v_jit > 0:
        gen = torch.Generator().manual_seed(11)
        for k in list(sd):
            if k.startswith("neuconw.") and k.endswith("weight_v"):
                sd[k] = sd[k] + v_jit * float(sd[k].abs().mean()) * torch.randn(sd[k].shape, generator=gen)
    for variance in (0.3, 0.5, 0.6, 0.7):
        sd["neuconw.deviation_network.variance"] = torch.tensor(float(variance))
        emb, neuconw, nerf, rdr = reference_system(sd)
        r, t, lb, c = rays[:R2], ts[:R2], label[:R2], rgbs[:R2]
        out = rdr.render(r.clone(), t, lb, perturb_overwrite=0, background_rgb=torch.zeros(1, 3), cos_anneal_ratio=0.5)
        out = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}  # (the reference's gradient() needs autograd)
        with torch.no_grad():
            sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
            o64 = O.render(sd64, cfg, r.double(), t, lb, 0.5, torch.zeros(1, 3, dtype=torch.float64))
            o32 = O.render(sd, cfg, r, t, lb, 0.5, torch.zeros(1, 3))
        row = {"variance": variance, "inv_s": float(torch.exp(torch.tensor(10.0 * variance))), "v_jit": v_jit, "rays": R2}
        for name, other in (("reference_fp32_vs_oracle_fp64", out), ("oracle_fp32_vs_oracle_fp64", o32)):
            row[name] = {k: float("%.3g" % rel(other[k], o64[k])) for k in ("color", "depth", "weights_sum", "weights")}
        floor.append(row)
        print(json.dumps(row))
res["fp32_noise_floor"] = floor

# ---- (3) the same comparison on the EXACT inputs of tests/test_gpu_fullsize.py::test_train_step_vs_oracle_at_trained_
# operating_points (tests/_parity.run_case: W = 256, 64 + 64, R = 16, seed 5, weight_g jitter 10 %, [weight_v jitter], the
# variance overridden), outputs AND parameter gradients: what the reference's own fp32 arithmetic differs from the fp64
# oracle by on the rays the GPU tolerances are quoted on
from tests._build import build_system, state_dict_cpu  # noqa: E402
from tests._parity import CFG as PCFG, perturb_weights  # noqa: E402
from tests._util import synth_rays  # noqa: E402

cfg_loss2 = G.AttrDict(NEUCONW=G.AttrDict(MESH_MASK_LIST=["sky"], DEPTH_LOSS=True, FLOOR_NORMAL=False))
ref_loss2 = ns.NeuconWLoss(coef=1.0, igr_weight=0.1, mask_weight=0.1, depth_weight=0.1, floor_weight=0.0, config=cfg_loss2)
pcfg = dict(PCFG, n_samples=64, n_importance=64)
at_tests = []
for variance, v_jit in ((0.5, 0.0), (0.6, 0.0), (0.7, 0.0), (0.6, 0.05)):
    e_, n_, f_, _ = build_system(W=256, n_a=48, n_vocab=100, nerf_w=256, color_hidden=256, head=128, seed=5, device="cpu", prec=0,
                                 n_samples=64, n_importance=64)
    perturb_weights(n_, 0.1, v_jit)
    with torch.no_grad():
        n_.deviation_network.variance.fill_(float(variance))
    sd = state_dict_cpu(e_, n_, f_, torch.float32)
    r, t, lb, c = synth_rays(16, 77, 100)
    emb, neuconw, nerf, rdr = reference_system(sd)
    out = rdr.render(r.clone(), t, lb, perturb_overwrite=0, background_rgb=torch.zeros(1, 3), cos_anneal_ratio=0.3)
    loss = sum(ref_loss2(out, c).values())
    named = {"embedding_a.weight": emb.weight}
    named.update({"neuconw." + k: v for k, v in neuconw.named_parameters() if not k.startswith("xyz_encoding_final")})
    named.update({"nerf." + k: v for k, v in nerf.named_parameters()})
    keys = list(named)
    g_ref = torch.autograd.grad(loss, [named[k] for k in keys], allow_unused=True)
    sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    o64 = O.render(sd64, pcfg, r.double(), t, lb, 0.3, torch.zeros(1, 3, dtype=torch.float64))
    l64 = O.neuconw_loss(o64, c.double(), pcfg)
    g64 = torch.autograd.grad(l64, [sd64[k] for k in keys], allow_unused=True)
    gmax, worst = {}, 0.0
    for k, g in zip(keys, g64):
        if g is not None:
            net = k.split(".")[1] if k.startswith("neuconw.") else k.split(".")[0]
            gmax[net] = max(gmax.get(net, 0.0), float(g.abs().max()))
    for k, a, b in zip(keys, g_ref, g64):
        if a is None or b is None:
            continue
        net = k.split(".")[1] if k.startswith("neuconw.") else k.split(".")[0]
        worst = max(worst, float((a.double() - b).abs().max()) / gmax[net])
    row = {"variance": variance, "v_jit": v_jit, "rays": 16,
           "reference_fp32_vs_oracle_fp64": {k: float("%.3g" % rel(out[k].detach(), o64[k].detach()))
                                             for k in ("color", "depth", "weights_sum", "weights", "gradient_error")},
           "param_grad_worst_rel_to_network_max": float("%.3g" % worst), "loss_reference": float(loss), "loss_oracle": float(l64)}
    at_tests.append(row)
    print(json.dumps(row))
res["at_test_inputs"] = at_tests
os.makedirs(os.path.dirname(args.out), exist_ok=True)
with open(args.out, "w") as f:
    json.dump(res, f, indent=1)
print("wrote", args.out)
