"""The forward-only render (the reference's validation / novel-view path, lightning_modules/neuconw_system.py:404-458 and
rendering/renderer.py:785-916 under torch.no_grad(); vertex colours, utils/visualization.py:138-150 -> renderer.rgb): the MLP
launches run their stash-free kernels (NULL stash members, include/neuconw_hip.h) -- outputs BITWISE equal to the training
forward in every precision and at every width, arenas reduced to the SDF network's h_l scratch + feat."""
import pytest
import torch

from tests._build import build_system
from tests._util import synth_rays

pytestmark = pytest.mark.gpu

BIG = dict(n_a=48, n_vocab=100, nerf_w=256, color_hidden=256, head=128)
KEYS = ("color", "color_sphere", "color_bg", "depth", "weights", "weights_sum", "weights_max", "cdf_fine", "gradients",
        "gradient_error", "mask_error", "inside_sphere", "s_val")


def _arena_bytes(mod):
    return {k: e["arena"].buf.numel() for k, e in mod.__dict__["_stash_cache"]._e.items()}


@pytest.mark.parametrize("W,prec_name,ns,ni,bg_dense", [(64, "f32", 8, 8, True), (64, "f16", 8, 8, False), (256, "f32", 16, 16, False),
                                                        (256, "f16", 64, 64, True), (256, "f16", 16, 16, False), (256, "bf16", 16, 16, True),
                                                        (512, "f16", 8, 16, False), (512, "bf16", 8, 16, True), (512, "f32", 8, 16, False)])
def test_no_grad_render_is_bitwise_the_training_forward(W, prec_name, ns, ni, bg_dense):
    import neuralrecon_w_amd as nw

    prec = {"f32": nw.PREC_F32, "bf16": nw.PREC_BF16, "f16": nw.PREC_F16}[prec_name]
    emb, neuconw, nerf, rdr = build_system(W=W, prec=prec, n_samples=ns, n_importance=ni, seed=3, **(BIG if W >= 256 else {}))
    rdr.bg_dense = bg_dense
    with torch.no_grad():
        for n, p in neuconw.named_parameters():
            if n.endswith("weight_g"):
                p.mul_(1.0 + 0.1 * torch.randn_like(p))
    R = 77  # ragged: not a multiple of the 32-point tiles / 128-point workgroups
    rays, ts, label, rgbs = [t.cuda() for t in synth_rays(R, 21, 64)]
    bg = torch.full((1, 3), 0.25).cuda()
    train = rdr.render(rays, ts, label, perturb_overwrite=0, background_rgb=bg, cos_anneal_ratio=0.4)
    assert train["color"].requires_grad
    with torch.no_grad():
        fwd = rdr.render(rays, ts, label, perturb_overwrite=0, background_rgb=bg, cos_anneal_ratio=0.4)
    assert not fwd["color"].requires_grad
    for k in KEYS:
        assert torch.equal(train[k].detach(), fwd[k]), k
    # what the forward-only render keeps in HBM: h_1 .. h_{L-1} + feat of the SDF network, nothing of the other two
    S = ns + ni
    esz = 4 if prec_name == "f32" else 2
    sizes = _arena_bytes(neuconw.sdf_net)
    k_train = next(k for k in sizes if k[-1] is True)
    k_fwd = next(k for k in sizes if k[-1] is False)
    tiles = ((R * S + 31) // 32 + 23) // 24 * 24
    h_lo = 8 if (W == 512 and prec_name == "f16") else 0  # (adj_mode 2: + the residuals of h for the adjoint sweep's phi', NcwSdfStash.s)
    assert sizes[k_fwd] == (9 + h_lo) * (W // 32) * tiles * 1024 * esz  # 8 hidden activations + feat
    assert sizes[k_train] > (2.3 if h_lo else 3.5) * sizes[k_fwd]        # (34 W / 32 + 6 blocks per tile against 9 W / 32; + 8 W / 32 on both sides with h_lo)
    for mod in (neuconw.color_net, nerf):
        sz = _arena_bytes(mod)
        assert max(v for k, v in sz.items() if k[-1] is False) <= 256, sz
    # the training forward is still intact afterwards: its backward runs and matches a fresh one
    loss = (train["color"] - rgbs).abs().mean() + 0.1 * train["gradient_error"].mean()
    loss.backward()
    g1 = neuconw.sdf_net.lin3.weight_v.grad.clone()
    for p_ in list(emb.parameters()) + list(neuconw.parameters()) + list(nerf.parameters()):
        p_.grad = None
    again = rdr.render(rays, ts, label, perturb_overwrite=0, background_rgb=bg, cos_anneal_ratio=0.4)
    ((again["color"] - rgbs).abs().mean() + 0.1 * again["gradient_error"].mean()).backward()
    g2 = neuconw.sdf_net.lin3.weight_v.grad
    assert torch.isfinite(g1).all() and float((g1 - g2).abs().max()) <= 2e-3 * float(g2.abs().max()) + 1e-12


def test_frozen_parameters_render_forward_only_and_backward_refuses():
    """No differentiable input (every parameter frozen): render() takes the forward-only kernels with grad mode ON as well."""
    import neuralrecon_w_amd as nw

    emb, neuconw, nerf, rdr = build_system(W=64, prec=nw.PREC_F32, n_samples=8, n_importance=8, seed=3)
    rays, ts, label, rgbs = [t.cuda() for t in synth_rays(40, 21, 64)]
    ref = rdr.render(rays, ts, label, perturb_overwrite=0, background_rgb=None, cos_anneal_ratio=0.4)
    for p_ in list(emb.parameters()) + list(neuconw.parameters()) + list(nerf.parameters()):
        p_.requires_grad_(False)
    out = rdr.render(rays, ts, label, perturb_overwrite=0, background_rgb=None, cos_anneal_ratio=0.4)
    assert not out["color"].requires_grad and torch.equal(out["color"], ref["color"].detach())


@pytest.mark.parametrize("W,prec_name", [(64, "f32"), (256, "f32"), (256, "f16"), (512, "f16")])
def test_inference_helpers_use_the_render_kernels(W, prec_name):
    """NeuconW.forward / renderer.rgb (vertex colours), NeuconW.gradient and NeRF.forward: stash-free, bitwise equal to the
    training launches on the same points."""
    import neuralrecon_w_amd as nw
    from neuralrecon_w_amd.neuconw import points_struct
    from neuralrecon_w_amd.stash import StashCache

    prec = {"f32": nw.PREC_F32, "f16": nw.PREC_F16}[prec_name]
    emb, neuconw, nerf, rdr = build_system(W=W, prec=prec, seed=4, **(BIG if W >= 256 else {}))
    rdr.infer_prec = prec
    n_a = emb.weight.shape[1]
    n = 1000
    g = torch.Generator().manual_seed(1)
    pts = (torch.rand(n, 3, generator=g) * 2 - 1).cuda()
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).cuda()
    a = torch.randn(n, n_a, generator=g).cuda()
    rgb = rdr.rgb(pts.unsqueeze(1), dirs.unsqueeze(1), a.unsqueeze(1))  # [N, 1, .] like utils/visualization.py:143-144
    grad = neuconw.gradient(pts, prec)
    ps = points_struct(x=pts, rays_d=dirs)
    sdf_t, grad_t, sctx = neuconw.sdf_net.fwd_stash(ps, n, prec)
    rgb_t, cctx = neuconw.color_net.fwd_stash(ps, n, prec, grad_t, a, sctx["arena"].ptr(sctx["ids"]["feat"]))
    assert torch.equal(grad, grad_t) and torch.equal(rgb, rgb_t.reshape(n, 3))
    StashCache.release(sctx["lease"]); StashCache.release(cctx["lease"])
    x4 = torch.cat([torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1), torch.rand(n, 1, generator=g)], -1).cuda()
    al, c = nerf(x4, dirs, a, prec)
    d_t, c_t, nctx = nerf.fwd_stash(points_struct(x=x4[:, :3].contiguous(), rays_d=dirs), n, prec, a, x4=x4)
    StashCache.release(nctx["lease"])
    assert torch.equal(al.reshape(-1), d_t) and torch.equal(c, c_t)
