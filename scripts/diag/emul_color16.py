"""CPU (no GPU): where does the fp16 mode's COLOUR error on TRAINED weights come from (1.9e-4 after 40 steps)?
The fp64 oracle is run with fp16 roundings injected at the places the kernels round (MFMA operands: weights and layer inputs;
f32 accumulate / epilogue), one group at a time:
  tail    the SDF network's feature rows + adjoint sweep (normals) in plain fp16 (the value chain stays exact = split path)
  cin     the colour network's INPUT operands (points, normals, view-dir encoding, features, appearance code) rounded
  clay    the colour network's hidden activations + all its weights rounded
The network is first trained for `--steps` Adam steps with the oracle itself (lr 1e-3, clip 0.99: tests/_parity.trained_weights).

    python scripts/diag/emul_color16.py [--steps 40 --R 16]
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
from tests._parity import CFG, perturb_weights  # noqa: E402
from tests._util import rel_err, synth_rays  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--R", type=int, default=16)
ap.add_argument("--variance", type=float, default=0.6)
ap.add_argument("--weights", default=None, help="a state_dict saved by the GPU (tests/_parity.trained_weights) instead of CPU training")
args = ap.parse_args()
cfg = dict(CFG, n_samples=64, n_importance=64)
H = torch.float16


def rnd(x, dt=H):
    return x.to(dt).to(x.dtype)


def split(x, dt=H):
    hi = rnd(x, dt)
    return hi + rnd(x - hi, dt)


MODE = {"adj_layers": None, "samp_w": None, "samp_h": None, "tail_feat": None, "tail_adj": None, "adj_t": None, "adj_w": None, "adj_s": None, "nin": None, "nda": None, "nw": None, "nact": None, "tail": None, "cin": None, "clay": None, "cw": None, "cin_f": None, "cin_da": None, "cin_p": None}   # None = exact; rnd / split


def q(x, key):
    f = MODE[key]
    if key.startswith("cin_") and f is None:
        f = MODE["cin"]
    if key.startswith("tail_") and f is None:
        f = MODE["tail"]
    if key in ("adj_t", "adj_w") and f is None:
        f = MODE["tail_adj"] if MODE["tail_adj"] is not None else MODE["tail"]
    return x if f is None else f(x)


def sdf_net_e(sd, x, prefix="sdf_net.", skip_in=(4,), multires=6, scale=1.0, with_grad=True):
    L = O._count_layers(sd, prefix)
    xs = x * scale
    gamma = O.freq_encode(xs, multires)
    h = gamma
    zs, Ws = [], []
    for l in range(L):
        W, b = O._lin_eff(sd, prefix + "lin%d" % l)
        if l in skip_in:
            h = torch.cat([h, gamma], 1) / math.sqrt(2.0)
        if l == L - 1:  # sdf row exact (split path), feature rows in the tail precision
            z = torch.cat([F.linear(h, W[:1], b[:1]), F.linear(q(h, "tail_feat"), q(W[1:], "tail_feat"), b[1:])], 1)
        elif not with_grad and (MODE["samp_w"] is not None or MODE["samp_h"] is not None):
            # the SAMPLER's queries (no gradient asked) with their own operand roundings: samp_w on the weights, samp_h on the layer input
            z = F.linear(q(h, "samp_h"), q(W, "samp_w"), b)
        else:
            z = F.linear(h, W, b)
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
            if MODE["adj_s"] is not None:  # phi'(z) recomputed from the STASHED (rounded) post-activation: 1 - exp(-100 h16(h))
                t = t * (1.0 - torch.exp(-100.0 * MODE["adj_s"](O.softplus100(zs[l]))))
            else:
                t = t * O.softplus100_d1(zs[l])
        if MODE.get("adj_layers") is not None and l not in MODE["adj_layers"]:  # round 6: the pairs only in some layers, single fp16 in the others
            qq = rnd(t) @ rnd(Ws[l])
        else:
            qq = q(t, "adj_t") @ q(Ws[l], "adj_w")
        if l in skip_in:
            qq = qq / math.sqrt(2.0)
            g_gamma = g_gamma + qq[:, -n_gamma:]
            qq = qq[:, :-n_gamma]
        t = qq
    g_gamma = g_gamma + t
    grad = O.freq_encode_jacobian_t_times(xs, multires, q(g_gamma, "tail_adj"))
    return sdf, feat, grad


def color_net_e(sd, points, normals, view_dirs, feat, a, prefix="color_net.", multires_view=4):
    d = O.freq_encode(view_dirs, multires_view)
    f = F.linear(q(feat, "cin_f"), q(sd[prefix + "xyz_encoding_final.weight"], "cw"), sd[prefix + "xyz_encoding_final.bias"])
    e = torch.cat([q(f, "clay"), q(d, "cin_da"), q(a, "cin_da")], 1)
    i = 0
    while (prefix + "static_encoding.static_linear_%d.weight" % i) in sd:
        e = F.relu(F.linear(e, q(sd[prefix + "static_encoding.static_linear_%d.weight" % i], "cw"),
                            sd[prefix + "static_encoding.static_linear_%d.bias" % i]))
        e = q(e, "clay")
        i += 1
    h = torch.cat([q(points, "cin_p"), q(normals, "cin_p"), e], -1)
    L = O._count_layers(sd, prefix)
    for l in range(L):
        W, b = O._lin_eff(sd, prefix + "lin%d" % l)
        h = F.linear(h, q(W, "cw"), b)
        if l < L - 1:
            h = q(F.relu(h), "clay")
    return torch.sigmoid(h)


def nerf_net_e(sd, pts, dirs, a, prefix="", skips=(4,), multires=10, multires_view=4):
    """background NeRF with roundings: nin = gamma(p) operand, nda = view-dir encoding + appearance code, nw = weights,
    nact = hidden activations"""
    gp = O.freq_encode(pts, multires)
    gv = O.freq_encode(dirs, multires_view)
    gpq = q(gp, "nin")
    h = gpq
    i = 0
    while (prefix + "pts_linears.%d.weight" % i) in sd:
        h = q(F.relu(F.linear(h, q(sd[prefix + "pts_linears.%d.weight" % i], "nw"), sd[prefix + "pts_linears.%d.bias" % i])), "nact")
        if i in skips:
            h = torch.cat([gpq, h], -1)
        i += 1
    density = F.linear(h, q(sd[prefix + "alpha_linear.weight"], "nw"), sd[prefix + "alpha_linear.bias"])
    feature = F.linear(h, q(sd[prefix + "feature_linear.weight"], "nw"), sd[prefix + "feature_linear.bias"])
    e = torch.cat([q(feature, "nact"), q(gv, "nda"), q(a, "nda")], -1)
    j = 0
    while (prefix + "apperence_encoding.static_linear_%d.weight" % j) in sd:
        e = q(F.relu(F.linear(e, q(sd[prefix + "apperence_encoding.static_linear_%d.weight" % j], "nw"),
                              sd[prefix + "apperence_encoding.static_linear_%d.bias" % j])), "nact")
        j += 1
    rgb = F.linear(e, q(sd[prefix + "rgb_linear.weight"], "nw"), sd[prefix + "rgb_linear.bias"])
    return density, rgb


def run(sd, rays, ts, label, emul):
    o_s, o_c, o_n = O.sdf_net, O.color_net, O.nerf_net
    if emul:
        O.sdf_net, O.color_net, O.nerf_net = sdf_net_e, color_net_e, nerf_net_e
    try:
        with torch.no_grad():
            return O.render(sd, cfg, rays.double(), ts, label, 0.3, torch.zeros(1, 3, dtype=torch.float64))
    finally:
        O.sdf_net, O.color_net, O.nerf_net = o_s, o_c, o_n


emb, neuconw, nerf, _ = build_system(W=256, n_a=48, n_vocab=100, nerf_w=256, color_hidden=256, head=128, seed=5, device="cpu", prec=0,
                                     n_samples=64, n_importance=64)
perturb_weights(neuconw, 0.1, 0.0)
sd = {k: v.float().requires_grad_(True) for k, v in state_dict_cpu(emb, neuconw, nerf, torch.float32).items()}
if args.weights:
    sd = {k: v.float() for k, v in torch.load(args.weights, map_location='cpu').items()}
elif args.steps > 0:  # tests/_parity.trained_weights: R = 128 rays (seed 123), lr 1e-3, eps 1e-7, clip 0.99, cos_anneal 0.3, perturb 0
    torch.set_num_threads(8)
    rays_t, ts_t, label_t, rgbs_t = synth_rays(128, 123, 100)
    opt = torch.optim.Adam(list(sd.values()), lr=1e-3, eps=1e-7)
    for i in range(args.steps):
        opt.zero_grad()
        out = O.render(sd, cfg, rays_t, ts_t, label_t, 0.3, torch.zeros(1, 3))
        loss = O.neuconw_loss(out, rgbs_t, cfg)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(list(sd.values()), 0.99)
        opt.step()
        if i % 10 == 0:
            print("train step %d loss %.5f" % (i, float(loss)), flush=True)
sd = {k: v.detach().double() for k, v in sd.items()}
sd["neuconw.deviation_network.variance"] = torch.tensor(args.variance, dtype=torch.float64)
rays, ts, label, rgbs = synth_rays(args.R, 77, 100)
ref = run(sd, rays, ts, label, False)
ALLC = {"tail": rnd, "cin": rnd, "cw": rnd, "clay": rnd}
ALLN = {"nin": rnd, "nda": rnd, "nw": rnd, "nact": rnd}
cases = [("exact (check)", {}), ("nerf all f16", ALLN), ("nerf dirs + a f16", {"nda": rnd}), ("nerf gamma(p) f16", {"nin": rnd}),
         ("nerf weights f16", {"nw": rnd}), ("everything incl. nerf f16", dict(ALLC, **ALLN)),
         ("everything f16, colour dirs+a exact", dict(ALLC, cin_da=(lambda x: x), **ALLN)),
         ("everything f16, colour AND nerf dirs+a exact", dict(ALLC, cin_da=(lambda x: x), **dict(ALLN, nda=None))), ("tail f16", {"tail": rnd}), ("colour inputs f16", {"cin": rnd}), ("colour weights f16", {"cw": rnd}),
         ("colour activations f16", {"clay": rnd}), ("colour net all f16", {"cin": rnd, "cw": rnd, "clay": rnd}),
         ("everything f16 (= the kernels)", {"tail": rnd, "cin": rnd, "cw": rnd, "clay": rnd}),
         ("tail f16, colour net split", {"tail": rnd, "cin": split, "cw": split, "clay": split}),
         ("tail split, colour net f16", {"tail": split, "cin": rnd, "cw": rnd, "clay": rnd}),
         ("colour: weights split, rest f16", {"tail": rnd, "cin": rnd, "cw": split, "clay": rnd}),
         ("colour: inputs split, rest f16", {"tail": rnd, "cin": split, "cw": rnd, "clay": rnd}),
         ("tail f16, cin+cw split, activations f16", {"tail": rnd, "cin": split, "cw": split, "clay": rnd}),
         ("only feat f16", {"cin_f": rnd}), ("only dirs + a f16", {"cin_da": rnd}), ("only points + normals f16", {"cin_p": rnd}),
         ("all f16 but points+normals split", {"tail": rnd, "cin": rnd, "cw": rnd, "clay": rnd, "cin_p": split}),
         ("all f16 but feat split", {"tail": rnd, "cin": rnd, "cw": rnd, "clay": rnd, "cin_f": split}),
         ("all f16 but points+normals+feat split", {"tail": rnd, "cin": rnd, "cw": rnd, "clay": rnd, "cin_p": split, "cin_f": split})]
for name, m in cases:
    for k in MODE:
        MODE[k] = m.get(k)
    out = run(sd, rays, ts, label, True)
    print("%-44s colour %.2e  depth %.2e  weights_sum %.2e  eikonal %.2e" % (
        name, rel_err(out["color"], ref["color"]), rel_err(out["depth"], ref["depth"]), rel_err(out["weights_sum"], ref["weights_sum"]),
        rel_err(out["gradient_error"], ref["gradient_error"])), flush=True)
