"""CPU (build container): the fp16 mode's colour error on the TIMED batch at the trained-weights parity point of bench.py
(`parity.trained_40_steps_inv_s_403`: 40 fp32 TrainSteps on the first 256 rays of the timed batch, variance 0.6), from the
state_dict the GPU run saved (`bench.py --save-trained-state`).  Two questions:
  (1) which fp16 rounding puts single rays above 1e-4 -- the kernels' roundings are injected into the fp64 oracle one
      candidate fix at a time (scripts/diag/emul_color16.py's emulation), per-ray distribution over the 256 rays;
  (2) how far the UNMODIFIED reference in fp32 is from the fp64 oracle on exactly these weights and rays (the "—" of DESIGN 4),
      when /root/reference is importable.

    python scripts/diag/emul_timed_batch.py gpurun_out/r05a/trained_state_bench.pt [--rays 256] [--no-reference]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
state_path = sys.argv[1]
n_rays = int(sys.argv[sys.argv.index("--rays") + 1]) if "--rays" in sys.argv else 256
with_ref = "--no-reference" not in sys.argv and "--only-new" not in sys.argv
argv_keep = list(sys.argv)
sys.argv = [sys.argv[0], "--steps", "0"]
src = open(os.path.join(ROOT, "scripts", "diag", "emul_color16.py")).read().split("emb, neuconw, nerf, _ = build_system")[0]
ns = {"__file__": os.path.join(ROOT, "scripts", "diag", "emul_color16.py"), "__name__": "defs"}
exec(compile(src, "defs", "exec"), ns)
O, rnd, split, MODE = ns["O"], ns["rnd"], ns["split"], ns["MODE"]
sys.argv = [argv_keep[0]]
import bench  # noqa: E402

torch.set_num_threads(8)
if "--shipped" in argv_keep:  # bench.py --config shipped: SDF 8 x 512, 8 + 16 samples (config/train_brandenburg_gate.yaml)
    bench.__dict__.update(W_SDF=512, N_SAMPLES=8, N_IMPORTANCE=16, M_SDF=2097664, M_SDF1=1835520, M_COL=585344)
batch_seed = int(argv_keep[argv_keep.index("--seed") + 1]) if "--seed" in argv_keep else 1000
sd0, cfg, (rays, ts, label, rgbs) = bench._oracle_setup(n_rays, batch_seed)
state = torch.load(state_path, map_location="cpu")
sd = {k: (v.double() if v.is_floating_point() else v) for k, v in state.items()}
print("variance in the saved state: %.3f (inv_s %.0f)" % (float(sd["neuconw.deviation_network.variance"]),
                                                          float(torch.exp(10 * sd["neuconw.deviation_network.variance"]))))


def run(modes):
    modes = dict(modes)
    tangent = modes.pop("tc", None)  # round 6: the normal's component ALONG THE RAY from a forward-mode tangent of the split value chain
    for k in MODE:
        MODE[k] = modes.get(k)
    keep = O.sdf_net, O.color_net, O.nerf_net, O.neuconw_forward
    if modes:
        O.sdf_net, O.color_net, O.nerf_net = ns["sdf_net_e"], ns["color_net_e"], ns["nerf_net_e"]
    if tangent is not None:
        exact_sdf = keep[0]

        def forward_tc(sd_, pts, dirs, a, cfg_, prefix=""):
            # the colour network sees the fp16-adjoint normals n16; what leaves for the compositor (true_cos = d . n,
            # rendering/renderer.py:613, and the eikonal term) has its component along d replaced by the tangent's
            sdf, feat, n16 = O.sdf_net(sd_, pts, prefix + "sdf_net.", cfg_.get("skip_in", (4,)), cfg_.get("multires", 6), cfg_.get("scale", 1.0))
            rgb = O.color_net(sd_, pts, n16, dirs, feat, a, prefix + "color_net.", cfg_.get("multires_view", 4))
            _, _, n_ex = exact_sdf(sd_, pts, prefix + "sdf_net.", cfg_.get("skip_in", (4,)), cfg_.get("multires", 6), cfg_.get("scale", 1.0))
            tc = tangent((dirs * n_ex).sum(-1, keepdim=True))  # (tangent = ident: exact; rnd32: an fp32-rounded scalar)
            n_out = n16 + dirs * (tc - (dirs * n16).sum(-1, keepdim=True)) / (dirs * dirs).sum(-1, keepdim=True)
            return rgb, O.inv_s_from_variance(sd_[prefix + "deviation_network.variance"]), sdf, n_out

        O.neuconw_forward = forward_tc
    try:
        with torch.no_grad():
            return O.render(sd, cfg, rays.double(), ts, label, 0.5, torch.zeros(1, 3, dtype=torch.float64))
    finally:
        O.sdf_net, O.color_net, O.nerf_net, O.neuconw_forward = keep


ref = run({})
scale = float(ref["color"].abs().max())


def per_ray(c):
    return (c["color"] - ref["color"]).abs().amax(-1) / scale


ident = lambda x: x  # noqa: E731
base = {"tail": rnd, "cin": rnd, "cw": split, "clay": rnd, "cin_da": ident, "nw": rnd, "nact": rnd, "nin": rnd, "nda": ident}
cases_all = [("the round-4 kernels (colour weights hi+lo, per-ray head columns fp32)", base),
         ("+ nerf weights hi+lo", dict(base, nw=split)),
         ("+ nerf gamma(p) hi+lo", dict(base, nin=split)),
         ("+ nerf weights + gamma(p) hi+lo", dict(base, nw=split, nin=split)),
         ("+ nerf everything hi+lo (weights, gamma(p), activations)", dict(base, nw=split, nin=split, nact=split)),
         ("+ colour inputs (feat, points, normals) hi+lo", dict(base, cin=split)),
         ("+ colour activations hi+lo", dict(base, clay=split)),
         ("+ colour inputs + activations hi+lo (whole colour net split)", dict(base, cin=split, clay=split)),
         ("+ SDF tail (feature rows + adjoint sweep) hi+lo", dict(base, tail=split)),
         ("+ feature rows only hi+lo (tail_feat)", dict(base, tail=None, tail_feat=split, tail_adj=rnd)),
         ("+ adjoint sweep only hi+lo (tail_adj: normals)", dict(base, tail=None, tail_feat=rnd, tail_adj=split)),
         ("+ adjoint hi+lo, phi' from the fp16 stash of h (the kernel's form)", dict(base, tail=None, tail_feat=rnd, tail_adj=split, adj_s=rnd)),
         ("+ adjoint: t hi+lo, W^T single (2 MFMAs), phi' from fp16 h", dict(base, tail=None, tail_feat=rnd, tail_adj=split, adj_w=rnd, adj_s=rnd)),
         ("+ adjoint: W^T hi+lo, t single (2 MFMAs), phi' from fp16 h", dict(base, tail=None, tail_feat=rnd, tail_adj=split, adj_t=rnd, adj_s=rnd)),
         ("+ kernel-form adjoint + nerf weights hi+lo", dict(base, tail=None, tail_feat=rnd, tail_adj=split, adj_s=rnd, nw=split)),
         ("+ kernel-form adjoint + nerf weights + gamma(p) hi+lo", dict(base, tail=None, tail_feat=rnd, tail_adj=split, adj_s=rnd, nw=split, nin=split)),
         ("+ kernel-form adjoint + nerf all hi+lo", dict(base, tail=None, tail_feat=rnd, tail_adj=split, adj_s=rnd, nw=split, nin=split, nact=split)),
         ("+ kernel-form adjoint + nerf all + colour inputs hi+lo", dict(base, tail=None, tail_feat=rnd, tail_adj=split, adj_s=rnd, nw=split, nin=split, nact=split, cin=split)),
         ("+ adjoint W^T hi+lo, t single, phi' from fp16 h (= round 5's kernels)", dict(base, tail=None, tail_feat=rnd, tail_adj=split, adj_t=rnd, adj_s=rnd)),
         ("+ adjoint W^T hi+lo, t single, phi' EXACT", dict(base, tail=None, tail_feat=rnd, tail_adj=split, adj_t=rnd)),
         ("+ adjoint W^T and t hi+lo, phi' EXACT", dict(base, tail=None, tail_feat=rnd, tail_adj=split)),
         ("+ tail + nerf weights hi+lo", dict(base, tail=split, nw=split)),
         ("+ tail + nerf gamma(p) hi+lo", dict(base, tail=split, nin=split)),
         ("+ tail + nerf activations hi+lo", dict(base, tail=split, nact=split)),
         ("+ tail + nerf weights + gamma(p) hi+lo", dict(base, tail=split, nw=split, nin=split)),
         ("+ tail + colour inputs hi+lo", dict(base, tail=split, cin=split)),
         ("+ tail + whole colour net hi+lo", dict(base, tail=split, cin=split, clay=split)),
         ("+ tail + colour net + nerf hi+lo (everything)", dict(base, tail=split, cin=split, clay=split, nw=split, nin=split, nact=split))]
r5 = dict(base, tail=None, tail_feat=rnd, tail_adj=split, adj_t=rnd, adj_s=rnd)  # round 5's kernels
if "--candidates" in argv_keep:
    cases_all = [("round-5 kernels (adjoint W^T hi+lo)", r5),
                 ("+ nerf weights hi+lo", dict(r5, nw=split)),
                 ("+ nerf weights + gamma(p) + activations hi+lo", dict(r5, nw=split, nin=split, nact=split)),
                 ("+ colour inputs (feat, points, normals) hi+lo", dict(r5, cin=split)),
                 ("+ colour input feat hi+lo only", dict(r5, cin_f=split)),
                 ("+ colour input points + normals hi+lo only", dict(r5, cin_p=split)),
                 ("+ nerf all + colour feat hi+lo", dict(r5, nw=split, nin=split, nact=split, cin_f=split)),
                 ("+ nerf all + colour points + normals hi+lo", dict(r5, nw=split, nin=split, nact=split, cin_p=split)),
                 ("+ nerf all + colour inputs hi+lo", dict(r5, nw=split, nin=split, nact=split, cin=split)),
                 ("+ nerf all + colour inputs + activations hi+lo", dict(r5, nw=split, nin=split, nact=split, cin=split, clay=split)),
                 ("+ colour activations hi+lo", dict(r5, clay=split)),
                 ("+ colour inputs + activations hi+lo", dict(r5, cin=split, clay=split)),
                 ("+ feature rows hi+lo", dict(r5, tail_feat=split)),
                 ("+ adjoint t hi+lo too", dict(r5, adj_t=None)),
                 ("+ nerf weights + colour inputs hi+lo", dict(r5, nw=split, cin=split)),
                 ("+ nerf weights + colour inputs + activations hi+lo", dict(r5, nw=split, cin=split, clay=split))]
now = dict(r5, nw=split, nin=split, nact=split, cin_p=split)  # the round's final kernels (R5.8)
if "--sampler" in argv_keep:  # the sampler's SDF queries (split precision, 3 MFMAs per product, as timed) with cheaper operand forms
    cases_all = [("final round-5 kernels, sampler in split precision (as shipped)", now),
                 ("sampler: weights hi+lo, layer inputs single fp16 (2 MFMAs)", dict(now, samp_h=rnd)),
                 ("sampler: layer inputs hi+lo, weights single fp16 (2 MFMAs)", dict(now, samp_w=rnd)),
                 ("sampler: plain fp16 (1 MFMA)", dict(now, samp_h=rnd, samp_w=rnd))]
if "--tangent" in argv_keep:  # round 6: true_cos from a forward-mode tangent, normals for the colour input stay fp16
    rnd32 = lambda x: x.float().double()  # noqa: E731
    w512 = dict(now, tail_adj=rnd, adj_t=rnd, adj_w=rnd)  # the shipped W = 512 kernels: adjoint sweep single-rounded (adj_split off)
    cases_all = [("round-5 kernels at W = 512 (adjoint sweep plain fp16)", w512),
                 ("+ true_cos from the tangent (exact), normals for the colour input fp16", dict(w512, tc=ident)),
                 ("+ true_cos from the tangent (fp32-rounded scalar)", dict(w512, tc=rnd32)),
                 ("+ tangent + adjoint W^T hi+lo (adj_split on)", dict(now, tc=ident)),
                 ("+ tangent + adjoint W^T and t hi+lo", dict(now, adj_t=None, tc=ident)),
                 ("+ tangent + colour activations hi+lo", dict(w512, clay=split, tc=ident)),
                 ("+ tangent + feature rows hi+lo", dict(w512, tail_feat=split, tc=ident)),
                 ("+ tangent + colour activations + feature rows + colour feat input hi+lo", dict(w512, clay=split, tail_feat=split, cin_f=split, tc=ident)),
                 ("adjoint W^T and t hi+lo, no tangent (round 5's emulated fix)", dict(now, adj_t=None)),
                 ("colour activations hi+lo, no tangent", dict(w512, clay=split)),
                 ("adjoint W^T and t hi+lo + colour activations hi+lo, no tangent", dict(now, adj_t=None, clay=split)),
                 ("adjoint W^T and t hi+lo, phi' exact + colour activations hi+lo, no tangent", dict(now, adj_t=None, adj_s=None, clay=split)),
                 ("adjoint W^T and t hi+lo + colour activations + colour feat input hi+lo", dict(now, adj_t=None, clay=split, cin_f=split)),
                 ("adjoint W^T and t hi+lo + colour activations + feature rows hi+lo", dict(now, adj_t=None, clay=split, tail_feat=split)),
                 ("+ tangent + colour activations hi+lo + adjoint W^T hi+lo", dict(now, clay=split, tc=ident))]
if "--adj-layers" in argv_keep:  # round 6: does the adjoint sweep need its operands as pairs in EVERY layer?
    full = dict(now, adj_t=None, clay=split)
    cases_all = [("adjoint W^T and t hi+lo in all layers + colour activations hi+lo (shipped, round 6)", full)] + [
        ("... pairs only in layers %s" % (sorted(ls),), dict(full, adj_layers=set(ls)))
        for ls in ([4, 5, 6, 7, 8], [5, 6, 7, 8], [3, 4, 5, 6, 7, 8], [4, 5, 6, 7], [0, 1, 2, 3, 4], [6, 7, 8], [0, 1, 2], [2, 3, 4, 5, 6], [1, 3, 5, 7], [0, 2, 4, 6, 8])]
if "--r6" in argv_keep:  # round 6: what is left at the shipped shape after the adjoint pairs + colour activation pairs
    cur = dict(now, adj_t=None, clay=split)
    cases_all = [("round-6 kernels (adjoint both operands as pairs, phi' from the fp16 stash of h; colour activations as pairs)", cur),
                 ("+ phi' exact in the adjoint sweep", dict(cur, adj_s=None)),
                 ("+ feature rows hi+lo", dict(cur, tail_feat=split)),
                 ("+ colour feat input hi+lo", dict(cur, cin_f=split)),
                 ("+ feature rows + colour feat input hi+lo", dict(cur, tail_feat=split, cin_f=split)),
                 ("+ phi' exact + feature rows + colour feat input hi+lo", dict(cur, adj_s=None, tail_feat=split, cin_f=split)),
                 ("+ final g_gamma exact (tail_adj None)", dict(cur, tail_adj=None)),
                 ("+ everything exact but the NeRF", dict(cur, adj_s=None, tail_feat=None, tail_adj=None, adj_w=None, cin_f=None, cin_p=None, clay=None, cw=None, cin_da=None, tail=None))]
if "--next" in argv_keep:  # what is left after R5.8, one candidate at a time
    cases_all = [("final round-5 kernels", now),
                 ("+ colour feat input hi+lo", dict(now, cin_f=split)),
                 ("+ colour activations hi+lo", dict(now, clay=split)),
                 ("+ colour feat + activations hi+lo (whole colour net split)", dict(now, cin_f=split, clay=split)),
                 ("+ feature rows hi+lo", dict(now, tail_feat=split)),
                 ("+ feature rows + colour feat input hi+lo", dict(now, tail_feat=split, cin_f=split)),
                 ("+ adjoint t hi+lo too", dict(now, adj_t=None)),
                 ("+ adjoint exact (t, W^T, phi')", dict(now, adj_t=None, adj_s=None)),
                 ("+ per-ray head columns exact instead of fp32 (cin_da)", dict(now)),
                 ("+ feature rows + colour feat + activations hi+lo", dict(now, tail_feat=split, cin_f=split, clay=split)),
                 ("+ everything hi+lo", dict(now, tail_feat=split, cin_f=split, clay=split, adj_t=None, adj_s=None))]
if "--only" in argv_keep:
    cases_all = [c for c in cases_all if argv_keep[argv_keep.index("--only") + 1] in c[0] or c is cases_all[0]]
cases = [c for c in cases_all if "--sampler" in argv_keep or "--next" in argv_keep or "--tangent" in argv_keep or "--adj-layers" in argv_keep or "--r6" in argv_keep or ("--only-new" not in argv_keep) or "--candidates" in argv_keep or ("kernel" in c[0] or "adjoint" in c[0] or "round-4" in c[0])]
res = {}
worst_rays = None
for name, m in cases:
    out = run(m)
    pr = per_ray(out)
    if worst_rays is None:
        worst_rays = torch.topk(pr, 6).indices.tolist()
        print("worst rays of the current kernels:", [(i, "%.2e" % float(pr[i])) for i in worst_rays])
        inside = out["inside_sphere"].double().mean(-1)
        print("  their weights_sum:", ["%.3f" % float(out["weights_sum"].reshape(-1)[i]) for i in worst_rays],
              " colour_bg share:", ["%.3f" % float(out["color_bg"][i].abs().max()) for i in worst_rays])
    res[name] = {"max": float(pr.max()), "p99": float(torch.quantile(pr, 0.99)), "rays_above_1e-4": int((pr > 1e-4).sum()),
                 "depth": float((out["depth"] - ref["depth"]).abs().max() / ref["depth"].abs().max())}
    print("%-66s max %.2e  p99 %.2e  rays > 1e-4: %d / %d   at the worst rays: %s" % (
        name, res[name]["max"], res[name]["p99"], res[name]["rays_above_1e-4"], n_rays,
        " ".join("%.1e" % float(pr[i]) for i in worst_rays[:4])), flush=True)

ref32 = None
if with_ref and os.path.isdir("/root/reference"):
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from oracle import ref_import  # noqa: E402
    import make_golden as G  # noqa: E402

    rns = ref_import.load()
    emb, neuconw, nerf, rdr = G.build_reference(rns, bench.W_SDF, 8, (4,), n_a=bench.N_A, n_vocab=state["embedding_a.weight"].shape[0], nerf_w=256,
                                                color_hidden=256, head=128, seed=0, n_samples=bench.N_SAMPLES, n_importance=bench.N_IMPORTANCE)
    with torch.no_grad():
        emb.weight.copy_(state["embedding_a.weight"])
        neuconw.load_state_dict({k[len("neuconw."):]: v for k, v in state.items() if k.startswith("neuconw.")}, strict=False)
        nerf.load_state_dict({k[len("nerf."):]: v for k, v in state.items() if k.startswith("nerf.")})
    # (grad mode on: the reference's SDFNetwork.gradient is an autograd.grad call)
    o32 = rdr.render(rays, ts, label, perturb_overwrite=0, background_rgb=torch.zeros(1, 3), cos_anneal_ratio=0.5)
    o32 = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in o32.items()}
    pr = (o32["color"].double() - ref["color"]).abs().amax(-1) / scale
    rel = lambda a, b: float((a.double().reshape(-1) - b.reshape(-1)).abs().max() / b.abs().max())  # noqa: E731
    ref32 = {"colour": rel(o32["color"], ref["color"]), "depth": rel(o32["depth"], ref["depth"]),
             "weights_sum": rel(o32["weights_sum"], ref["weights_sum"]), "weights": rel(o32["weights"], ref["weights"]),
             "colour_p99": float(torch.quantile(pr, 0.99)), "rays_above_1e-4": int((pr > 1e-4).sum())}
    print("the UNMODIFIED reference in fp32 vs the fp64 oracle on these weights / rays:", {k: ("%.2e" % v if isinstance(v, float) else v) for k, v in ref32.items()})
out_path = os.path.join(ROOT, "profiles", "r06" if ("--tangent" in argv_keep or "--adj-layers" in argv_keep or "--r6" in argv_keep) else "r05", "emul_timed_batch%s%s%s.json" % (("_shipped" if "--shipped" in argv_keep else "") + ("_tangent" if "--tangent" in argv_keep else "") + ("_adj_layers" if "--adj-layers" in argv_keep else "") + ("_r6" if "--r6" in argv_keep else ""),
                                                                                "_kernel_form" if "--only-new" in argv_keep else "",
                                                                                "" if batch_seed == 1000 else "_seed%d" % batch_seed))
os.makedirs(os.path.dirname(out_path), exist_ok=True)
with open(out_path, "w") as fh:
    json.dump({"state": os.path.relpath(os.path.abspath(state_path), ROOT), "rays": n_rays, "emulated_fp16_candidates": res,
               "reference_fp32_vs_fp64_oracle_trained_40_steps": ref32}, fh, indent=1)
print("wrote", out_path)
