"""GPU: dead-background elimination (ncw_bg_select + NcwPoints mode 4 + NcwWgradDesc.n_points_dev).  With trim_sphere the
compositor multiplies the background NeRF's output of a primary sample inside the unit sphere by 1 - inside_sphere = 0
(/root/reference rendering/renderer.py:637,693-708), forward and backward, so every mode (16-bit and the fp32 parity mode)
evaluates the NeRF only where it can matter.  Nothing observable may change: rendered outputs bitwise, parameter gradients
to summation order (the weight-gradient K-slices cover different point sets)."""
import pytest
import torch

from tests._build import build_system, loss_from_outputs, named_params
from tests._util import rel_err, synth_rays

pytestmark = pytest.mark.gpu


def _rays_crossing_the_sphere(R, seed):
    """rays whose sampled interval [near, far] leaves the unit sphere on both ends for many samples"""
    rays, ts, label, rgbs = synth_rays(R, seed, 64)
    rays = rays.clone()
    rays[:, 6] = 0.6   # near: |o + d z| ~ 1.4 -> outside
    rays[:, 7] = 3.6   # far:  ~1.6 -> outside
    return rays, ts, label, rgbs


def test_selection_list_matches_torch():
    from neuralrecon_w_amd import lib as L

    R, S, O = 1500, 37, 4  # more rays than one pass of the kernel's workgroup (1024)
    g = torch.Generator().manual_seed(3)
    o = torch.tensor([0.0, 0.0, -2.0]) + 0.1 * torch.randn(R, 3, generator=g)
    d = torch.nn.functional.normalize(torch.tensor([0.0, 0.0, 1.0]) + 0.1 * torch.randn(R, 3, generator=g), dim=-1)
    z = torch.sort(0.6 + 3.0 * torch.rand(R, S, generator=g), dim=-1).values
    sd = torch.full((R,), 3.0 / S)
    from oracle import neuconw_oracle as Or

    need = Or.bg_needed(o, d, z, sd, O)  # pinned to the reference's inside_sphere (tests/test_reference_bg_is_dead.py)
    inside = ~need[:, :S]
    want = torch.nonzero(need.reshape(-1)).reshape(-1).int()
    oc, dc, zc, sc = o.cuda(), d.cuda(), z.cuda().contiguous(), sd.cuda()  # the PRIMARY z: columns pair by index
    idx = torch.full((R * (S + O),), -1, dtype=torch.int32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    offs = torch.empty(R + 1, dtype=torch.int32, device="cuda")
    L.check(L.get_lib().ncw_bg_select(L.ptr(oc), L.ptr(dc), L.ptr(zc), L.ptr(sc), R, S, O, L.ptr(idx), L.ptr(offs), L.ptr(cnt),
                                      L.stream_ptr(oc.device)), "ncw_bg_select")
    n = int(cnt)
    assert int(offs[0]) == 0 and int(offs[R]) == n and bool((offs[1:] - offs[:-1] >= O).all())
    # a mid-point within rounding of |p| = 1 may fall either way between torch's norm and the kernel's sqrt: allow a few
    got = idx[:n].cpu()
    sym = set(got.tolist()) ^ set(want.tolist())
    assert len(sym) <= 4, (n, want.numel(), sorted(sym)[:10])
    assert bool((got[1:] > got[:-1]).all()), "ray-major, ascending"
    assert 0.05 < float(inside.float().mean()) < 0.95  # the case exercises both branches


@pytest.mark.parametrize("prec_name", ["f16", "bf16", "f32"])
def test_elimination_changes_nothing(prec_name):
    """f32: the generic weights-through-LDS NeRF kernels in mode 4, per-point appearance-code rows at the ray sample's slot,
    order-fixed split-K over the tiles that exist (the parity mode stays bitwise run-to-run reproducible: test_gpu_repro.py)."""
    import neuralrecon_w_amd as nw

    prec = {"f16": nw.PREC_F16, "bf16": nw.PREC_BF16, "f32": nw.PREC_F32}[prec_name]
    res = {}
    for dense in (True, False):
        emb, neuconw, nerf, rdr = build_system(W=256, n_a=48, n_vocab=64, nerf_w=256, color_hidden=256, head=128, seed=5,
                                               prec=prec, n_samples=32, n_importance=32)
        rdr.bg_dense = dense
        rays, ts, label, rgbs = _rays_crossing_the_sphere(96, 7)
        out = rdr.render(rays.cuda(), ts.cuda(), label.cuda(), perturb_overwrite=0, background_rgb=torch.zeros(1, 3).cuda(),
                         cos_anneal_ratio=0.3)
        loss = loss_from_outputs(out, rgbs.cuda())
        loss.backward()
        res[dense] = (out, float(loss), {k: p.grad.detach().clone() for k, p in named_params(emb, neuconw, nerf).items()
                                         if p.grad is not None})
    (od, ld, gd), (oe, le, ge) = res[True], res[False]
    ins = od["inside_sphere"]
    frac = float(ins.float().mean())
    assert 0.1 < frac < 0.95, frac  # both kinds of primary samples present
    for k in ("color", "depth", "weights_sum", "weights", "color_bg", "gradient_error"):
        assert torch.equal(od[k], oe[k]), (k, float((od[k] - oe[k]).abs().max()))
    assert ld == le
    worst = max(rel_err(ge[k], gd[k]) for k in gd)
    print("%s: inside fraction %.2f, worst parameter-gradient difference dense vs eliminated %.2e" % (prec_name, frac, worst))
    assert set(gd) == set(ge) and worst < 2e-5, worst  # f32 atomics of the weight-gradient slices: summation order only
    if prec_name == "f32":  # nothing of the SDF / colour networks changes at all; only the background sums re-associate
        for k in gd:
            if not k.startswith("nerf.") and not k.startswith("embedding"):
                assert torch.equal(gd[k], ge[k]), k


def test_selection_is_refused_elsewhere():
    """mode-4 points reach the background kernels only: the SDF entry points refuse them loudly"""
    import neuralrecon_w_amd as nw
    from neuralrecon_w_amd import lib as L
    from neuralrecon_w_amd.neuconw import points_struct
    from tests.test_gpu_sdf import _mk

    net = _mk(64, 8, (4,))
    idx = torch.zeros(64, dtype=torch.int32, device="cuda")
    cnt = torch.ones(1, dtype=torch.int32, device="cuda")
    z = torch.rand(2, 32, device="cuda")
    pts = points_struct(rays_o=torch.zeros(2, 3, device="cuda"), rays_d=torch.ones(2, 3, device="cuda"), z=z,
                        sample_dist=torch.ones(2, device="cuda"), mode=2)
    pts4 = points_struct(mode=4, idx=idx, count=cnt)
    pts4.rays_o, pts4.rays_d, pts4.z, pts4.sample_dist, pts4.per_ray = pts.rays_o, pts.rays_d, pts.z, pts.sample_dist, 32
    plan = net.packed(nw.PREC_BF16)
    out = torch.empty(64, device="cuda")
    rc = L.get_lib().ncw_sdf_infer_points(plan.net, nw.PREC_BF16, pts4, 64, L.ptr(out), L.stream_ptr(out.device))
    assert rc == -2  # NCW_E_UNSUPPORTED


def test_elimination_in_the_small_generic_kernels():
    """W = 64 background net (generic kernels in every precision), fp32 and fp16: dense vs eliminated."""
    import neuralrecon_w_amd as nw

    for prec in (nw.PREC_F32, nw.PREC_F16):
        res = {}
        for dense in (True, False):
            emb, neuconw, nerf, rdr = build_system(seed=2, prec=prec)
            rdr.bg_dense = dense
            rays, ts, label, rgbs = _rays_crossing_the_sphere(70, 9)
            out = rdr.render(rays.cuda(), ts.cuda(), label.cuda(), perturb_overwrite=0, background_rgb=torch.zeros(1, 3).cuda(),
                             cos_anneal_ratio=0.3)
            loss_from_outputs(out, rgbs.cuda()).backward()
            res[dense] = (out, {k: p.grad.detach().clone() for k, p in named_params(emb, neuconw, nerf).items() if p.grad is not None})
        (od, gd), (oe, ge) = res[True], res[False]
        for k in ("color", "depth", "weights", "color_bg"):
            assert torch.equal(od[k], oe[k]), (prec, k)
        worst = max(rel_err(ge[k], gd[k]) for k in gd)
        assert worst < 2e-5, (prec, worst)


def test_elimination_under_graph_capture():
    """TrainStep(capture=True) in the fp16 mode: the selection, the device-sized launches and the zero-filled dense outputs
    inside a HIP graph replay like the eager step."""
    import neuralrecon_w_amd as nw

    R, steps = 64, 6
    rays, ts, label, rgbs = [t.cuda() for t in _rays_crossing_the_sphere(R, 12)]
    bg = torch.zeros(1, 3, device="cuda")
    curves = []
    for capture in (False, True):
        emb, neuconw, nerf, rdr = build_system(W=256, n_a=48, n_vocab=64, nerf_w=256, color_hidden=256, head=128, seed=6,
                                               prec=nw.PREC_F16, n_samples=16, n_importance=16)
        rdr.sync_free = True
        assert nerf.supports_selection(nw.PREC_F16) and not rdr.bg_dense
        train = nw.TrainStep(rdr, [emb, neuconw, nerf], loss_from_outputs, lr=1e-3, eps=1e-7, clip=0.99, capture=capture,
                             capture_warmup=3)
        curves.append([float(train(rays, ts, label, rgbs, background_rgb=bg, cos_anneal_ratio=0.1 * i, perturb_overwrite=0)[0])
                       for i in range(steps)])
        if capture:
            assert train._graphs is not None
    for a, b in zip(*curves):
        assert abs(a - b) <= 2e-3 * max(1.0, abs(a)), curves  # 16-bit runs separate through Adam's sign steps
    assert curves[0][-1] < curves[0][0]


@pytest.mark.parametrize("dense", [True, False])
def test_background_refinement_in_split_precision(dense):
    """fp16 mode, W = 256: ncw_nerf_refine re-evaluates the samples the compositor can use (ncw_bg_select's list) with gamma_10(p4),
    weights and activations as fp16 hi + lo pairs over the plain-fp16 outputs of ncw_nerf_fwd (models/nerf.py:156-182 on the points of
    rendering/renderer.py:176-186).  At the selected samples density / raw rgb must be at the fp32 level of the fp64 oracle (the plain
    kernel: 3-9e-4); the other samples keep the plain values (dense) or zeros (elimination); `.refine = False` gives the plain outputs."""
    import neuralrecon_w_amd as nw
    from neuralrecon_w_amd.neuconw import points_struct
    from neuralrecon_w_amd.stash import StashCache
    from neuralrecon_w_amd import rayops
    from oracle import neuconw_oracle as O
    from tests._build import build_system, state_dict_cpu
    from tests._util import rel_err, synth_rays

    emb, neuconw, nerf, rdr = build_system(W=256, n_a=48, n_vocab=100, nerf_w=256, color_hidden=256, head=128, seed=7, prec=nw.PREC_F16,
                                           n_samples=16, n_importance=16)
    with torch.no_grad():
        for p_ in nerf.parameters():
            p_.add_(0.02 * torch.randn_like(p_))
    R, S, O_ = 50, 32, 4
    rays, ts, label, _ = synth_rays(R, 31, 100)
    rays_o, rays_d = rays[:, 0:3].cuda().contiguous(), rays[:, 3:6].cuda().contiguous()
    g = torch.Generator().manual_seed(3)
    z = torch.sort(1.0 + 2.5 * torch.rand(R, S, generator=g), -1)[0].cuda()      # primary samples: some leave the unit sphere
    z_out = torch.sort(3.6 + 3.0 * torch.rand(R, O_, generator=g), -1)[0].cuda()
    sample_dist = torch.full((R, 1), 0.05).cuda()
    z_feed, _ = rayops.sort_merge(z, z_out)
    M = S + O_
    a = emb.weight.detach()[ts.cuda()].contiguous()
    pts = points_struct(rays_o=rays_o, rays_d=rays_d, z=z_feed, sample_dist=sample_dist, mode=2)
    sel = (z, O_)
    outs = {}
    for refine in (True, False):
        nerf.refine = refine
        nerf._plans.clear()  # (the residual matrices belong to the pack plan: built once per precision)
        den, rgb, c = nerf.fwd_stash(pts, R * M, nw.PREC_F16, a, select=None if dense else sel, train=False, refine=sel)
        StashCache.release(c["lease"])
        outs[refine] = (den.view(R, M).cpu(), rgb.view(R, M, 3).cpu())
        if refine:
            idx = c["lease"]["sel_idx"][: int(c["lease"]["sel_count"])].cpu().long()
    sd = {k[len("nerf."):]: v for k, v in state_dict_cpu(emb, neuconw, nerf, torch.float64).items() if k.startswith("nerf.")}
    rgb_r, _, den_r = O.render_core_outside({"nerf." + k: v for k, v in sd.items()}, rays_o.cpu().double(), rays_d.cpu().double(),
                                            z_feed.cpu().double(), sample_dist.cpu().double(), a.cpu().double(), "nerf.")
    keep = torch.zeros(R * M, dtype=torch.bool)
    keep[idx] = True
    keep = keep.view(R, M)
    assert 0.08 < float(keep.float().mean()) < 0.9 and bool(keep[:, S:].all())
    e = {}
    for refine in (True, False):
        den, rgb = outs[refine]
        e[refine] = (rel_err(den[keep], den_r[keep]), rel_err(rgb[keep], rgb_r[keep]))
    print("background NeRF at the %d selected samples (dense=%s): split-precision refinement density %.2e rgb %.2e; plain fp16 %.2e / %.2e"
          % (int(keep.sum()), dense, e[True][0], e[True][1], e[False][0], e[False][1]))
    assert max(e[True]) < 2e-5 and min(e[False]) > 20 * max(e[True]), e   # (measured 7.6e-6 / 1.3e-6 against 5.5e-4 / 7.3e-4)
    # what the refinement does NOT touch
    if dense:
        assert torch.equal(outs[True][0][~keep], outs[False][0][~keep]) and torch.equal(outs[True][1][~keep], outs[False][1][~keep])
    else:
        assert float(outs[True][0][~keep].abs().max()) == 0.0 and float(outs[True][1][~keep].abs().max()) == 0.0
