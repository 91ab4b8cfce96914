"""GPU unit parity of the colour network (models/neuconw.py:59-170) and the background NeRF (8a-6,
models/nerf.py:86-183) on their own: forward, backward (input adjoints) and every weight gradient of ncw_color_fwd/bwd,
ncw_nerf_fwd/bwd + ncw_wgrad against the fp64 oracle's autograd, and the forward against outputs of the real reference
(tests/golden/units_w64.npz).  The end-to-end tests (test_gpu_render.py, test_gpu_fullsize.py) see these kernels only
through the compositor."""
import pytest
import torch

from tests._build import build_system, load_golden_weights
from tests._util import load_golden, rel_err

pytestmark = pytest.mark.gpu

# Outputs: max-norm relative error (the north-star measure).  Adjoints and weight gradients of these ReLU networks:
# relative L2 (Frobenius) error, because a pre-activation that rounds across 0 flips its ReLU mask and changes THAT
# point's adjoint by O(1): in the max norm over per-point adjoints one flip shows as 0.1-0.27 (both 16-bit types, 4e6 unit
# evaluations here) and even fp32 has one or two (1.7e-2 at W = 256); the L2 error then grows like sqrt(eps), which is
# why fp16 is only ~2.8x better than bf16 on this measure.  Bounds = about 2x the errors measured on MI355X (printed):
#   fp32 <= 1e-7 outputs, <= 6e-4 L2;  fp16 9e-4 outputs, 3.2e-2 adjoints, 4.7e-2 weight gradients;  bf16 6e-3, 9e-2, 0.13
TOL = {"f32": dict(out=1e-4, adj=2e-3, grad=2e-3), "f16": dict(out=2e-3, adj=6e-2, grad=0.1),
       "bf16": dict(out=1.5e-2, adj=0.2, grad=0.27)}


def fro_err(a, b):
    a, b = a.detach().double().reshape(-1), b.detach().double().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-300))


def _err(prec_name):
    return fro_err


def flipped_points(a, b, tol=1e-4):
    """rows (points) of a per-point adjoint whose max error exceeds tol x the tensor's max -- in fp32 these are the points
    where one pre-activation rounded across 0 against the fp64 oracle; every other row must agree to the fp32 bar."""
    a, b = a.detach().double(), b.detach().double()
    return int(((a - b).abs().amax(dim=-1) > tol * b.abs().max()).sum())


def _prec(name):
    import neuralrecon_w_amd as nw

    return {"f32": nw.PREC_F32, "bf16": nw.PREC_BF16, "f16": nw.PREC_F16}[name]


def _jitter(mod, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in mod.named_parameters():
            if n.endswith("weight_g"):
                p.mul_(1.0 + 0.1 * torch.randn(p.shape, generator=g).to(p.device))
            if n.endswith("bias"):
                p.add_(0.05 * torch.randn(p.shape, generator=g).to(p.device))


def _wgrads(mod, ctx, prec, n):
    from neuralrecon_w_amd.stash import WgradBatch

    plan = ctx["plan"]
    plan.g_arena.zero_()
    batch = WgradBatch(plan.g_arena.device, prec, n)
    mod.add_wgrads(ctx, batch)
    batch.run()
    grads = {id(p): torch.zeros_like(p) for p in mod.parameters()}
    keep = plan.unpack_grads(grads)
    torch.cuda.synchronize()
    return {k: grads[id(p)].cpu() for k, p in mod.named_parameters()}, keep


def _unit(n, seed):
    g = torch.Generator().manual_seed(seed)
    d = torch.randn(n, 3, generator=g)
    return d / d.norm(dim=-1, keepdim=True)


@pytest.mark.parametrize("W,n_a,head", [(64, 16, 32), (256, 48, 128)])
@pytest.mark.parametrize("prec_name", ["f32", "f16", "bf16"])
def test_color_net_train_vs_oracle(W, n_a, head, prec_name):
    from neuralrecon_w_amd.neuconw import points_struct
    from neuralrecon_w_amd.stash import StashArena
    from oracle import neuconw_oracle as O

    prec = _prec(prec_name)
    _, neuconw, _, _ = build_system(W=W, n_a=n_a, color_hidden=W, head=head, nerf_w=64, seed=11, prec=prec)
    cn = neuconw.color_net
    _jitter(cn, 1)
    n = 4000  # ragged: not a multiple of 128
    g = torch.Generator().manual_seed(2)
    x = (torch.rand(n, 3, generator=g) * 2 - 1) * 0.9
    normals = torch.randn(n, 3, generator=g)
    dirs = _unit(n, 3)
    feat = 0.5 * torch.randn(n, W, generator=g)
    a = torch.randn(n, n_a, generator=g)
    w_rgb = torch.randn(n, 3, generator=g)
    # ---- oracle, fp64 ------------------------------------------------------------------------------------------
    sd = {"color_net." + k: v.detach().cpu().double().requires_grad_(True) for k, v in cn.state_dict().items()}
    ins = [t.double().requires_grad_(True) for t in (normals, feat, a)]
    rgb_r = O.color_net(sd, x.double(), ins[0], dirs.double(), ins[1], ins[2])
    names = list(sd)
    gr = torch.autograd.grad((rgb_r * w_rgb.double()).sum(), [sd[k] for k in names] + ins)
    gref, (dn_r, df_r, da_r) = dict(zip(names, gr[:len(names)])), gr[len(names):]
    # ---- HIP ---------------------------------------------------------------------------------------------------
    dev = torch.device("cuda")
    pts = points_struct(x=x.to(dev), rays_d=dirs.to(dev))
    ar = StashArena(dev, prec, n)
    fid, dfid = ar.new(W // 32), ar.new(W // 32)
    ar.allocate(zero=True)
    ar.from_rows(fid, feat.to(dev))
    rgb, ctx = cn.fwd_stash(pts, n, prec, normals.to(dev), a.to(dev), ar.ptr(fid))
    d_grad = torch.zeros(n, 3, device=dev)
    d_a = torch.zeros(n, n_a, device=dev)
    cn.bwd_stash(ctx, w_rgb.to(dev), d_grad, d_a, ar.ptr(dfid))
    d_feat = ar.to_rows(dfid, W)
    got, _keep = _wgrads(cn, ctx, prec, n)
    t = TOL[prec_name]
    e_out = rel_err(rgb.cpu(), rgb_r)
    err = _err(prec_name)
    adj = dict(normals=err(d_grad.cpu(), dn_r), feat=err(d_feat.cpu(), df_r), a=err(d_a.cpu(), da_r))
    e_adj = max(adj.values())
    per = {k: err(got[k], gref["color_net." + k]) for k in got}
    e_grad = max(per.values())
    print("color W=%d %s: rgb %.2e, input adjoints %s, weight gradients %.2e (worst: %s)"
          % (W, prec_name, e_out, {k: "%.2e" % v for k, v in adj.items()}, e_grad, max(per, key=per.get)))
    if prec_name == "f32":  # the L2 error above is a handful of ReLU flips: all other points sit at the 1e-4 bar
        nflip = max(flipped_points(d_grad.cpu(), dn_r), flipped_points(d_feat.cpu(), df_r), flipped_points(d_a.cpu(), da_r))
        print("   fp32: %d of %d points beyond 1e-4 (ReLU mask flips against fp64)" % (nflip, n))
        assert nflip <= n // 200
    assert e_out < t["out"] and e_adj < t["adj"] and e_grad < t["grad"], (e_out, e_adj, e_grad)


@pytest.mark.parametrize("W,n_a", [(64, 16), (256, 48)])
@pytest.mark.parametrize("prec_name", ["f32", "f16", "bf16"])
def test_nerf_train_vs_oracle(W, n_a, prec_name):
    from neuralrecon_w_amd.neuconw import points_struct
    from oracle import neuconw_oracle as O

    prec = _prec(prec_name)
    _, _, nerf, _ = build_system(W=64, n_a=n_a, nerf_w=W, seed=12, prec=prec)
    _jitter(nerf, 4)
    n = 4000
    g = torch.Generator().manual_seed(5)
    p3 = _unit(n, 6)
    inv_r = torch.rand(n, 1, generator=g) * 0.9 + 0.05  # the inverted-sphere point [x / r, 1 / r] (renderer.py:181-186)
    x4 = torch.cat([p3, inv_r], -1)
    dirs = _unit(n, 7)
    a = torch.randn(n, n_a, generator=g)
    w_den, w_rgb = torch.randn(n, generator=g), torch.randn(n, 3, generator=g)
    sd = {k: v.detach().cpu().double().requires_grad_(True) for k, v in nerf.state_dict().items()}
    a_r = a.double().requires_grad_(True)
    den_r, rgb_r = O.nerf_net(sd, x4.double(), dirs.double(), a_r)
    names = list(sd)
    gr = torch.autograd.grad((den_r[:, 0] * w_den.double()).sum() + (rgb_r * w_rgb.double()).sum(),
                             [sd[k] for k in names] + [a_r], allow_unused=True)
    gref, da_r = {k: v for k, v in zip(names, gr[:-1]) if v is not None}, gr[-1]
    dev = torch.device("cuda")
    pts = points_struct(x=x4[:, :3].contiguous().to(dev), rays_d=dirs.to(dev))
    density, rgb, ctx = nerf.fwd_stash(pts, n, prec, a.to(dev), x4=x4.to(dev))
    d_a = torch.zeros(n, n_a, device=dev)
    nerf.bwd_stash(ctx, w_den.to(dev), w_rgb.to(dev), d_a)
    got, _keep = _wgrads(nerf, ctx, prec, n)
    t = TOL[prec_name]
    e_out = max(rel_err(density.cpu(), den_r[:, 0]), rel_err(rgb.cpu(), rgb_r))
    err = _err(prec_name)
    e_adj = err(d_a.cpu(), da_r)
    per = {k: err(got[k], gref[k]) for k in got if k in gref}
    e_grad = max(per.values())
    print("nerf W=%d %s: density/rgb %.2e, d_a %.2e, weight gradients %.2e (worst: %s)"
          % (W, prec_name, e_out, e_adj, e_grad, max(per, key=per.get)))
    if prec_name == "f32":
        nflip = flipped_points(d_a.cpu(), da_r)
        print("   fp32: %d of %d points beyond 1e-4 (ReLU mask flips against fp64)" % (nflip, n))
        assert nflip <= n // 200
    assert e_out < t["out"] and e_adj < t["adj"] and e_grad < t["grad"], (e_out, e_adj, e_grad)


@pytest.mark.parametrize("prec_name", ["f32", "f16"])
def test_color_and_nerf_forward_reference_golden(prec_name):
    """Weights, inputs and outputs straight from the real reference (tests/golden/make_golden.py -> units_w64.npz):
    RenderingNetwork.forward(points, normals, view_dirs, feature_vectors, a) and NeRF.forward(p4, dirs, a)."""
    from neuralrecon_w_amd.neuconw import points_struct
    from neuralrecon_w_amd.stash import StashArena, StashCache

    prec = _prec(prec_name)
    sd, _, _, m = load_golden("units_w64")
    emb, neuconw, nerf, _ = build_system(W=64, n_a=16, nerf_w=64, color_hidden=64, head=32, seed=0, prec=prec)
    load_golden_weights(sd, emb, neuconw, nerf)
    dev = torch.device("cuda")
    n = m["x"].shape[0]
    pts = points_struct(x=m["x"].to(dev), rays_d=m["dirs"].to(dev))
    ar = StashArena(dev, prec, n)
    fid = ar.new(2)
    ar.allocate(zero=True)
    ar.from_rows(fid, m["feat"].to(dev))
    rgb, ctx = neuconw.color_net.fwd_stash(pts, n, prec, m["grad"].to(dev), m["a"].to(dev), ar.ptr(fid))
    StashCache.release(ctx["lease"])
    alpha, bg_rgb = nerf(m["p4"].to(dev), m["dirs"].to(dev), m["a"].to(dev), prec=prec)
    tol = 1e-4 if prec_name == "f32" else 2e-3
    e = (rel_err(rgb.cpu(), m["rgb"]), rel_err(alpha.cpu(), m["density"]), rel_err(bg_rgb.cpu(), m["bg_rgb"]))
    print("golden units %s: colour rgb %.2e, nerf density %.2e, nerf rgb %.2e" % ((prec_name,) + e))
    assert max(e) < tol, e


@pytest.mark.parametrize("prec_name", ["f16", "bf16"])
def test_color_head_per_ray_bias(prec_name):
    """16-bit modes: the view-direction / appearance-code columns of the head's first layer are evaluated once per ray in fp32
    (ncw_aux_ray_bias, models/neuconw.py:131-140) instead of as 16-bit MFMA operands.  Checked here: the per-ray rows against the
    fp64 product (fp32 accuracy), and that the forward with them agrees with the 16-bit-operand forward and the oracle within
    the unit tolerance.  What it buys shows on TRAINED weights, where the appearance code matters: colour error of the composed
    step 1.88e-4 -> 1.02e-4 (tests/test_gpu_fullsize.py::test_train_step_vs_oracle_after_training, scripts/diag/emul_color16.py)."""
    from neuralrecon_w_amd.neuconw import points_struct
    from neuralrecon_w_amd.stash import StashArena
    from oracle import neuconw_oracle as O

    W, n_a, head = 256, 48, 128
    prec = _prec(prec_name)
    _, neuconw, _, _ = build_system(W=W, n_a=n_a, color_hidden=W, head=head, nerf_w=64, seed=11, prec=prec)
    cn = neuconw.color_net
    _jitter(cn, 1)
    with torch.no_grad():
        cn.static_encoding[0].weight[:, W:] *= 12.0  # a head that leans on the appearance code like a trained one
    R, S = 64, 32
    n = R * S
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(n, 3, generator=g) * 2 - 1) * 0.9
    normals = torch.randn(n, 3, generator=g)
    dirs_r = _unit(R, 3)
    a_r = torch.randn(R, n_a, generator=g)
    feat = 0.5 * torch.randn(n, W, generator=g)
    sd = {"color_net." + k: v.detach().cpu().double() for k, v in cn.state_dict().items()}
    rgb_ref = O.color_net(sd, x.double(), normals.double(), dirs_r.repeat_interleave(S, 0).double(), feat.double(),
                          a_r.repeat_interleave(S, 0).double())
    dev = torch.device("cuda")
    pts = points_struct(x=x.to(dev), rays_d=dirs_r.repeat_interleave(S, 0).to(dev))
    ar = StashArena(dev, prec, n)
    fid = ar.new(W // 32)
    ar.allocate(zero=True)
    ar.from_rows(fid, feat.to(dev))
    out = {}
    for on in (False, True):
        cn.ray_bias = on
        rgb, ctx = cn.fwd_stash(pts, n, prec, normals.to(dev), a_r.repeat_interleave(S, 0).to(dev), ar.ptr(fid))
        out[on] = rgb.cpu()
        assert (ctx["stash"].aux_bias is not None) == on
        if on:  # the rows themselves: W_e0[:, W:] . [gamma_4(d) | a] per "ray" (here: per point row)
            rows = ctx["lease"]["aux_bias"].cpu().double()
            w0 = cn.static_encoding[0].weight.detach().cpu().double()[:, W:]
            xin = torch.cat([O.freq_encode(dirs_r.double(), 4), a_r.double()], 1).repeat_interleave(S, 0)
            want = xin @ w0.t()
            e_rows = rel_err(rows, want)
            print("ncw_aux_ray_bias rows vs fp64: %.2e" % e_rows)
            assert e_rows < 2e-6
    e_on, e_off = rel_err(out[True], rgb_ref), rel_err(out[False], rgb_ref)
    print("colour forward %s: per-ray fp32 bias %.2e, 16-bit operands %.2e" % (prec_name, e_on, e_off))
    assert e_on < TOL[prec_name]["out"] and e_off < TOL[prec_name]["out"]
    assert e_on < 1.5 * e_off + 1e-5


@pytest.mark.parametrize("W", [256, 512])
def test_color_forward_with_split_activations(W):
    """Round 6, fp16 mode: `RenderingNetwork.act_split` (NcwColorNet.act_split; the default at d_feature = 256 / 512) --
    the activations of every layer as fp16 hi + lo pairs on top of the hi + lo weights: a third pass W_hi x_lo of the weight ring per
    layer (csrc/ncw_color.hip SPLIT = 2).  What is left single-rounded is the feature vector it reads from the stash and the per-ray
    view / appearance columns (fp32).  rgb against the fp64 oracle must drop well below the weights-only form's, the stash (the
    backward's operands: single-rounded) and therefore the backward must not move, and the forward-only render is the same bit for bit."""
    from neuralrecon_w_amd.neuconw import points_struct
    from neuralrecon_w_amd.stash import StashArena
    from oracle import neuconw_oracle as O

    import neuralrecon_w_amd as nw

    n_a, head = 48, 128
    prec = nw.PREC_F16
    _, neuconw, _, _ = build_system(W=W, n_a=n_a, color_hidden=256, head=head, nerf_w=64, seed=11, prec=prec)
    cn = neuconw.color_net
    _jitter(cn, 1)
    assert cn.plan(prec).net.act_split == 1  # the default at both shipped widths
    R, S = 125, 32  # ragged
    n = R * S
    g = torch.Generator().manual_seed(2)
    x = (torch.rand(n, 3, generator=g) * 2 - 1) * 0.9
    normals = torch.randn(n, 3, generator=g)
    dirs = _unit(R, 3).repeat_interleave(S, 0)
    a = torch.randn(R, n_a, generator=g).repeat_interleave(S, 0)
    feat = (0.5 * torch.randn(n, W, generator=g)).half().float()  # exactly representable: the stash holds fp16
    sd = {"color_net." + k: v.detach().cpu().double() for k, v in cn.state_dict().items()}
    ref = O.color_net(sd, x.double(), normals.double(), dirs.double(), feat.double(), a.double())
    dev = torch.device("cuda")
    pts = points_struct(x=x.to(dev), rays_d=dirs.to(dev))
    ar = StashArena(dev, prec, n)
    fid = ar.new(W // 32)
    ar.allocate(zero=True)
    ar.from_rows(fid, feat.to(dev))
    out, stash = {}, {}
    for on in (False, True):
        cn.act_split = on
        rgb, ctx = cn.fwd_stash(pts, n, prec, normals.to(dev), a.to(dev), ar.ptr(fid))
        assert ctx["plan"].net.act_split == int(on)
        out[on] = rgb.cpu()
        stash[on] = ctx["arena"].to_rows(ctx["ids"]["x"][1], 256).cpu()
    e_off, e_on = rel_err(out[False], ref), rel_err(out[True], ref)
    print("colour forward W=%d fp16: weights as pairs %.2e, weights + activations as pairs %.2e" % (W, e_off, e_on))
    assert e_on < 0.25 * e_off and e_on < 2e-5, (e_off, e_on)
    assert rel_err(stash[True], stash[False]) < 2e-3  # the same single-rounded stash (up to what the more accurate inputs change)
    with torch.no_grad():
        rgb_r, _ = cn.fwd_stash(pts, n, prec, normals.to(dev), a.to(dev), ar.ptr(fid), train=False)
    assert torch.equal(rgb_r.cpu(), out[True])
