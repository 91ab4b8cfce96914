"""GPU: the per-step glue launches (ray prologue, variance network, NeuconWLoss, weights_max, embedding backward) against
the torch expressions of the reference they replace (renderer.py:793-806, models/neuconw.py:173-179 + renderer.py:624-632,
losses.py:21-43, renderer.py:905, nn.Embedding)."""
import ctypes as C

import pytest
import torch

from tests._build import build_system, loss_from_outputs, named_params
from tests._util import synth_rays

pytestmark = pytest.mark.gpu


def test_ray_prologue_matches_torch():
    from neuralrecon_w_amd import lib as L

    R = 777
    for ncols in (8, 11):
        rays = torch.randn(R, ncols, device="cuda")
        origin, radius = torch.tensor([0.3, -0.2, 0.1]), 2.4
        out = [torch.empty(R, 3, device="cuda"), torch.empty(R, 3, device="cuda")] + [torch.empty(R, device="cuda") for _ in range(4)]
        oh = (C.c_float * 3)(*origin.tolist())
        L.check(L.get_lib().ncw_ray_prologue(L.ptr(rays), ncols, R, oh, radius, *[L.ptr(t) for t in out], L.stream_ptr(rays.device)), "p")
        o = origin.cuda()
        assert torch.equal(out[0], ((rays[:, 0:3] - o).float() / radius).float())
        assert torch.equal(out[1], rays[:, 3:6])
        assert torch.equal(out[2], rays[:, 6] / radius) and torch.equal(out[3], rays[:, 7] / radius)
        if ncols >= 10:
            assert torch.equal(out[4], rays[:, 8] / radius) and torch.equal(out[5], rays[:, 9])
        else:
            assert not out[4].any() and not out[5].any()


@pytest.mark.parametrize("v", [0.3, 0.03, -2.0, 1.5])
def test_inv_s_forward_backward(v):
    from neuralrecon_w_amd import lib as L

    var = torch.tensor(v, device="cuda", requires_grad=True)
    inv_s, s_val = torch.empty(1, device="cuda"), torch.empty(1, device="cuda")
    L.check(L.get_lib().ncw_inv_s_fwd(L.ptr(var.detach().reshape(1)), L.ptr(inv_s), L.ptr(s_val), L.stream_ptr(var.device)), "f")
    ref = torch.exp(var * 10.0).clamp(1e-6, 1e6)
    assert torch.allclose(inv_s, ref.detach().reshape(1), rtol=2e-6) and torch.allclose(s_val, 1.0 / inv_s)
    d = torch.randn(1000, device="cuda")
    (ref * d.sum()).backward()
    d_var = torch.empty(1, device="cuda")
    L.check(L.get_lib().ncw_inv_s_bwd(L.ptr(d), 1000, L.ptr(inv_s), L.ptr(d_var), L.stream_ptr(var.device)), "b")
    assert torch.allclose(d_var, var.grad.reshape(1), rtol=1e-4, atol=1e-6 * float(var.grad.abs()) + 1e-12)
    again = torch.empty(1, device="cuda")
    L.check(L.get_lib().ncw_inv_s_bwd(L.ptr(d), 1000, L.ptr(inv_s), L.ptr(again), L.stream_ptr(var.device)), "b")
    assert torch.equal(d_var, again)  # fixed-order reduction


@pytest.mark.parametrize("use_mask,use_depth,n_sfm", [(True, True, None), (False, True, 37), (True, False, None), (False, False, None)])
def test_fused_loss_matches_torch(use_mask, use_depth, n_sfm):
    import neuralrecon_w_amd as nw

    R = 513
    g = torch.Generator(device="cuda").manual_seed(0)
    mk = lambda *s: torch.rand(*s, device="cuda", generator=g).requires_grad_(True)  # noqa: E731
    out = {"color": mk(R, 3), "gradient_error": mk(1), "mask_error": mk(R, 1), "sfm_depth_loss": mk(n_sfm or R)}
    rgbs = torch.rand(R, 3, device="cuda", generator=g)
    rgbs[5] = out["color"][5].detach()  # exact zeros of the L1 term: sgn(0) = 0 like torch
    loss_mod = nw.NeuconWLoss(coef=1.5, igr_weight=1e-4, mask_weight=0.1, depth_weight=0.2, use_mask=use_mask, use_depth=use_depth)
    ref = sum(loss_mod.terms(out, rgbs).values())
    gref = torch.autograd.grad(ref, [v for v in out.values()], allow_unused=True)
    got = loss_mod(out, rgbs)
    assert got.shape == () and abs(float(got) - float(ref)) <= 2e-6 * abs(float(ref))
    ggot = torch.autograd.grad(got * 1.0, [v for v in out.values()], allow_unused=True)
    for k, a, b in zip(out, ggot, gref):
        assert (a is None) == (b is None), k
        if a is not None:
            assert a.shape == b.shape and torch.allclose(a, b, rtol=2e-6, atol=1e-12), k


def test_render_outputs_and_training_step_unchanged_by_the_glue_launches():
    """weights_max, s_val and the embedding gradient of the fused path against torch on the same render; the flat trainer
    (direct embedding-gradient accumulation) against autograd's accumulation."""
    import neuralrecon_w_amd as nw

    emb, neuconw, nerf, rdr = build_system(seed=2, prec=nw.PREC_BF16)
    rays, ts, label, rgbs = [t.cuda() for t in synth_rays(96, seed=5, n_vocab=64)]
    ts[:40] = ts[0]  # many repeated images: the case torch's embedding backward serialises
    out = rdr.render(rays, ts, label, perturb_overwrite=0, background_rgb=torch.zeros(1, 3, device="cuda"), cos_anneal_ratio=0.2)
    assert torch.equal(out["weights_max"], out["weights"].max(dim=-1, keepdim=True)[0])
    inv_s = torch.exp(neuconw.deviation_network.variance * 10.0).clamp(1e-6, 1e6)
    assert torch.allclose(out["s_val"], (1.0 / inv_s).reshape(1, 1), rtol=2e-6)
    loss_from_outputs(out, rgbs).backward()
    g_fused = emb.weight.grad.detach().clone()
    emb.weight.grad = None
    rdr.reproducible = True  # torch's embedding + ordered reductions
    out2 = rdr.render(rays, ts, label, perturb_overwrite=0, background_rgb=torch.zeros(1, 3, device="cuda"), cos_anneal_ratio=0.2)
    loss_from_outputs(out2, rgbs).backward()
    assert float((g_fused - emb.weight.grad).abs().max()) <= 2e-5 * float(emb.weight.grad.abs().max()) + 1e-9
    # flat trainer: embedding gradient accumulated straight into the flat buffer
    emb3, neuconw3, nerf3, rdr3 = build_system(seed=2, prec=nw.PREC_BF16)
    train = nw.TrainStep(rdr3, [emb3, neuconw3, nerf3], loss_from_outputs, lr=0.0, eps=1e-7, clip=None)
    train(rays, ts, label, rgbs, background_rgb=torch.zeros(1, 3, device="cuda"), cos_anneal_ratio=0.2, perturb_overwrite=0)
    off, k = train.fp.slices[id(emb3.weight)]
    g_flat = train.fp.flat_grad[off:off + k].view_as(emb3.weight)
    assert float((g_flat - g_fused).abs().max()) <= 2e-5 * float(g_fused.abs().max()) + 1e-9
    assert named_params(emb3, neuconw3, nerf3)["embedding_a.weight"].grad.data_ptr() == g_flat.data_ptr()
