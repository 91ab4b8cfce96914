"""CPU emulation of the 16-bit SDF network at trained operating points (no GPU): which part of the SDF chain has to be
more accurate than one 16-bit rounding per operand for the rendered outputs to stay within 1e-4 of the fp64 oracle when
inv_s = exp(10 variance) is in the hundreds.  The oracle's sdf_net is replaced by a version that rounds weights and
activations (and optionally keeps a hi + lo split of either) the way the MFMA kernels do; everything else stays fp64.

    python scripts/diag/emul16.py [--ns 64 --ni 64 --R 16]
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
from tests._util import rel_err, synth_rays  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ns", type=int, default=64)
ap.add_argument("--ni", type=int, default=64)
ap.add_argument("--R", type=int, default=16)
ap.add_argument("--W", type=int, default=256)
ap.add_argument("--vjit", type=float, default=0.05)
args = ap.parse_args()

CFG = dict(n_outside=4, up_sample_steps=2, s_val_base=3, render_bg=True, trim_sphere=True, mesh_mask_list=["sky"],
           depth_loss=True, igr_weight=0.1, mask_weight=0.1, depth_weight=0.1, skip_in=(4,), multires=6, multires_view=4,
           n_samples=args.ns, n_importance=args.ni)


def rnd(x, dt):
    return x.to(dt).to(x.dtype)


def make_sdf_net(dt, split_w=(), split_h=(), f32_epi=True):
    """sdf_net with operands rounded to `dt`; layers in split_w / split_h keep hi + lo parts of W / h."""
    def sdf_net(sd, x, prefix="sdf_net.", skip_in=(4,), multires=6, scale=1.0, with_grad=True):
        L = O._count_layers(sd, prefix)
        xs = x * scale
        gamma = O.freq_encode(xs, multires)
        h = gamma
        zs, Ws = [], []
        for l in range(L):
            W, b = O._lin_eff(sd, prefix + "lin%d" % l)
            if l in skip_in:
                h = torch.cat([h, gamma], 1) / math.sqrt(2.0)
            Wr = rnd(W, dt)
            if l in split_w:
                Wr = Wr + rnd(W - Wr, dt)
            hr = rnd(h, dt)
            if l in split_h:
                hr = hr + rnd(h - hr, dt)
            z = F.linear(hr, Wr, b)
            if f32_epi:
                z = z.float().to(z.dtype)
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
            q = rnd(t, dt) @ rnd(Ws[l], dt)
            if l in skip_in:
                q = q / math.sqrt(2.0)
                g_gamma = g_gamma + q[:, -n_gamma:]
                q = q[:, :-n_gamma]
            t = q
        g_gamma = g_gamma + t
        grad = O.freq_encode_jacobian_t_times(xs, multires, g_gamma)
        return sdf, feat, grad
    return sdf_net


def run(sd, rays, ts, label, sdf_fn=None, dtype=torch.float64):
    orig = O.sdf_net
    if sdf_fn is not None:
        O.sdf_net = sdf_fn
    try:
        sdd = {k: v.to(dtype) for k, v in sd.items()}
        with torch.no_grad():
            return O.render(sdd, CFG, rays.to(dtype), ts, label, 0.3, torch.zeros(1, 3, dtype=dtype))
    finally:
        O.sdf_net = orig


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
L = 9
variants = [("fp32 oracle", None, torch.float32),
            ("f16", make_sdf_net(torch.float16), torch.float64),
            ("f16 splitW all", make_sdf_net(torch.float16, split_w=range(L)), torch.float64),
            ("f16 splitH all", make_sdf_net(torch.float16, split_h=range(L)), torch.float64),
            ("f16 splitWH all", make_sdf_net(torch.float16, split_w=range(L), split_h=range(L)), torch.float64),
            ("f16 splitWH last1", make_sdf_net(torch.float16, split_w=[L - 1], split_h=[L - 1]), torch.float64),
            ("f16 splitWH last3", make_sdf_net(torch.float16, split_w=range(L - 3, L), split_h=range(L - 3, L)), torch.float64),
            ("bf16", make_sdf_net(torch.bfloat16), torch.float64)]
for var in (0.3, 0.5, 0.6, 0.7):
    sd = state_dict_cpu(emb, neuconw, nerf, torch.float64)
    sd["neuconw.deviation_network.variance"] = torch.tensor(var, dtype=torch.float64)
    ref = run(sd, rays, ts, label)
    print("variance %.1f (inv_s %.0f):" % (var, math.exp(10 * var)))
    for name, fn, dt in variants:
        out = run(sd, rays, ts, label, fn, dt)
        e = {k: rel_err(out[k], ref[k]) for k in ("color", "depth", "weights_sum", "weights", "gradient_error")}
        print("   %-20s " % name + "  ".join("%s %.1e" % (k, v) for k, v in e.items()))
