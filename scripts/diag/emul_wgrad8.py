"""CPU study (no GPU): what an 8-bit copy of the WEIGHT-GRADIENT-ONLY stash operands (the cotangents zbar_l / qbar_l of the
SDF network, the pre-activation cotangents of the colour / background networks) costs in parameter-gradient accuracy.
The fp64 oracle is run with every Linear replaced by an autograd function whose weight gradient is formed from quantised
operands (data gradients stay exact), i.e. exactly what a wgrad launch reading 8-bit stashes would compute.

    python scripts/diag/emul_wgrad8.py [--R 16 --ns 64 --ni 64 --variance 0.3]
"""
import argparse
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import neuconw_oracle as O  # noqa: E402
from tests._build import build_system, state_dict_cpu  # noqa: E402
from tests._util import synth_rays  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ns", type=int, default=64)
ap.add_argument("--ni", type=int, default=64)
ap.add_argument("--R", type=int, default=16)
ap.add_argument("--W", type=int, default=256)
ap.add_argument("--variance", type=float, default=0.3)
ap.add_argument("--vjit", type=float, default=0.05)
ap.add_argument("--short", action="store_true", help="only the modes of the R-scaling study (NOTEBOOK R5.1)")
args = ap.parse_args()

CFG = dict(n_outside=4, up_sample_steps=2, s_val_base=3, render_bg=True, trim_sphere=True, mesh_mask_list=["sky"],
           depth_loss=True, igr_weight=0.1, mask_weight=0.1, depth_weight=0.1, skip_in=(4,), multires=6, multires_view=4,
           n_samples=args.ns, n_importance=args.ni)

MODE = {"x": "f16", "y": "f16"}


def _tiles(v, tp, tf):
    """[N, Fd] -> padded view [N/tp, tp, Fd/tf, tf] for per-tile scales."""
    N, Fd = v.shape
    Np, Fp = -(-N // tp) * tp, -(-Fd // tf) * tf
    w = torch.zeros(Np, Fp, dtype=v.dtype)
    w[:N, :Fd] = v
    return w.view(Np // tp, tp, Fp // tf, tf), N, Fd


def q8(v, fmt, tp, tf):
    """8-bit quantisation with one power-of-two scale per (tp points x tf features) tile."""
    dt, fmax = (torch.float8_e4m3fn, 448.0) if fmt == "e4m3" else (torch.float8_e5m2, 57344.0)
    w, N, Fd = _tiles(v, tp, tf)
    amax = w.abs().amax(dim=(1, 3), keepdim=True).clamp_min(1e-300)
    sc = torch.exp2(torch.floor(torch.log2(fmax / amax)))
    r = (w * sc).float().clamp(-fmax, fmax).to(dt).to(v.dtype) / sc
    return r.reshape(w.shape[0] * tp, -1)[:N, :Fd]


def qi8(v, tp, tf):
    """int8 with one f32 scale per (tp points x tf features) tile (absolute error ~ amax / 254)."""
    w, N, Fd = _tiles(v, tp, tf)
    amax = w.abs().amax(dim=(1, 3), keepdim=True).clamp_min(1e-300)
    sc = 127.0 / amax
    r = torch.round(w * sc).clamp(-127, 127) / sc
    return r.reshape(w.shape[0] * tp, -1)[:N, :Fd]


def quant(v, mode):
    if mode == "exact":
        return v
    if mode == "f16":
        return v.to(torch.float16).to(v.dtype)
    if mode == "bf16":
        return v.to(torch.bfloat16).to(v.dtype)
    fmt, tp, tf = mode.split(":")
    if fmt == "i8":
        return qi8(v, int(tp), int(tf))
    return q8(v, fmt, int(tp), int(tf))


class LinQ(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        ctx.has_b = b is not None
        y = x @ W.t()
        return y + b if b is not None else y

    @staticmethod
    def backward(ctx, gy):
        x, W = ctx.saved_tensors
        gx = gy @ W
        gyq = quant(gy, MODE["y"])
        gW = gyq.t() @ quant(x, MODE["x"])
        return gx, gW, (gyq.sum(0) if ctx.has_b else None)


def linear_q(x, W, b=None):
    return LinQ.apply(x, W, b)


def sdf_net_q(sd, x, prefix="sdf_net.", skip_in=(4,), multires=6, scale=1.0, with_grad=True):
    L = O._count_layers(sd, prefix)
    xs = x * scale
    gamma = O.freq_encode(xs, multires)
    h = gamma
    zs, Ws = [], []
    for l in range(L):
        W, b = O._lin_eff(sd, prefix + "lin%d" % l)
        if l in skip_in:
            h = torch.cat([h, gamma], 1) / math.sqrt(2.0)
        z = linear_q(h, W, b)
        zs.append(z)
        Ws.append(W)
        h = O.softplus100(z) if l < L - 1 else z
    sdf = h[:, 0] / scale
    feat = h[:, 1:]
    if not with_grad:
        return sdf, feat, None
    n_gamma = gamma.shape[1]
    g_gamma = torch.zeros_like(gamma)
    t = torch.zeros_like(zs[-1])
    t[:, 0] = 1.0
    for l in range(L - 1, -1, -1):
        if l < L - 1:
            t = t * O.softplus100_d1(zs[l])
        q = linear_q(t, Ws[l].t())  # q = t @ W: the weight gradient of this "Linear" is qbar^T t  (the second-order product)
        if l in skip_in:
            q = q / math.sqrt(2.0)
            g_gamma = g_gamma + q[:, -n_gamma:]
            q = q[:, :-n_gamma]
        t = q
    g_gamma = g_gamma + t
    grad = O.freq_encode_jacobian_t_times(xs, multires, g_gamma)
    return sdf, feat, grad


def grads(sd0, rays, ts, label, rgbs, patched):
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd0.items()}
    orig_sdf, orig_lin = O.sdf_net, F.linear
    if patched:
        O.sdf_net = sdf_net_q
        F.linear = linear_q
    try:
        out = O.render(sd, CFG, rays.double(), ts, label, 0.3, torch.zeros(1, 3, dtype=torch.float64))
        loss = O.neuconw_loss(out, rgbs.double(), CFG)
        names = [k for k, v in sd.items() if v.requires_grad]
        gs = torch.autograd.grad(loss, [sd[k] for k in names], allow_unused=True)
    finally:
        O.sdf_net, F.linear = orig_sdf, orig_lin
    return {k: (g if g is not None else torch.zeros_like(sd[k])) for k, g in zip(names, gs)}


def net_of(k):
    for n in ("sdf_net", "color_net", "nerf"):
        if n in k:
            return n
    return "other"


emb, neuconw, nerf, rdr = build_system(W=args.W, n_a=48, n_vocab=100, nerf_w=256, color_hidden=256, head=128, seed=5,
                                       device="cpu", prec=0, n_samples=args.ns, n_importance=args.ni)
torch.manual_seed(11)
with torch.no_grad():
    for n, p in neuconw.named_parameters():
        if n.endswith("weight_g"):
            p.mul_(1.0 + 0.1 * torch.randn_like(p))
        elif n.endswith("weight_v") and args.vjit > 0:
            p.add_(args.vjit * p.abs().mean() * torch.randn_like(p))
rays, ts, label, rgbs = synth_rays(args.R, 77, 100)
sd0 = state_dict_cpu(emb, neuconw, nerf, torch.float64)
sd0["neuconw.deviation_network.variance"] = torch.tensor(args.variance, dtype=torch.float64)
ref = grads(sd0, rays, ts, label, rgbs, False)
gmax = {}
for k, g in ref.items():
    gmax[net_of(k)] = max(gmax.get(net_of(k), 0.0), float(g.abs().max()))
print("R %d  %d+%d  variance %.1f; largest gradient per network:" % (args.R, args.ns, args.ni, args.variance), gmax)
variants = [("exact", "exact"), ("f16", "f16"), ("f16", "bf16"), ("f16", "e4m3:32:32"), ("f16", "e5m2:32:32"), ("e4m3:32:32", "e4m3:32:32"),
            ("f16", "i8:32:1"), ("f16", "i8:32:32"), ("f16", "i8:1:32"), ("f16", "i8:1:256"), ("f16", "i8:4:4"), ("i8:1:32", "i8:1:32")]
if args.short:
    variants = [("f16", "f16"), ("bf16", "bf16"), ("f16", "e4m3:32:32"), ("e4m3:32:32", "e4m3:32:32"), ("e5m2:32:32", "e5m2:32:32"), ("i8:1:32", "i8:1:32")]
for mx, my in variants:
    MODE["x"], MODE["y"] = mx, my
    g = grads(sd0, rays, ts, label, rgbs, True)
    worst = {}
    l2 = {}
    for k in ref:
        n = net_of(k)
        e = float((g[k] - ref[k]).abs().max()) / gmax[n]
        if e > worst.get(n, (0, ""))[0]:
            worst[n] = (e, k)
        l2.setdefault(n, [0.0, 0.0])
        l2[n][0] += float(((g[k] - ref[k]) ** 2).sum())
        l2[n][1] += float((ref[k] ** 2).sum())
    print("x %-11s y %-11s " % (mx, my) + "  ".join("%s max %.1e (%s) l2 %.1e" % (n, worst[n][0], worst[n][1].split(".")[-2] + "." + worst[n][1].split(".")[-1],
                                                                          math.sqrt(l2[n][0] / l2[n][1])) for n in sorted(worst)))
