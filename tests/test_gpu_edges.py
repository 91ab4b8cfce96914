"""Edge cases of the hot path through the C ABI: empty inputs, a single ray, the maximum samples-per-ray the
per-ray kernels hold in LDS (S + O = 512 in the standard object, 1088 in the large-ray one) and past it (must fail loudly), bad arguments."""
import pytest
import torch

from tests._build import build_system, loss_from_outputs, state_dict_cpu
from tests._util import rel_err, synth_rays

pytestmark = pytest.mark.gpu

CFG = dict(n_outside=4, up_sample_steps=2, s_val_base=3, render_bg=True, trim_sphere=True, mesh_mask_list=["sky"],
           depth_loss=True, igr_weight=0.1, mask_weight=0.1, depth_weight=0.1, skip_in=(4,), multires=6, multires_view=4)


def test_empty_point_sets():
    import neuralrecon_w_amd as nw
    from neuralrecon_w_amd import grid

    emb, neuconw, nerf, rdr = build_system(seed=1, prec=nw.PREC_F32)
    out = rdr.sdf(torch.zeros(0, 3, device="cuda"))
    assert out.shape[0] == 0
    got = grid.sdf_grid_range(neuconw.sdf_net, 8, (-1, -1, -1), (1, 1, 1), start=5, count=0)
    assert got.numel() == 0
    torch.cuda.synchronize()


def test_single_ray_matches_oracle():
    import neuralrecon_w_amd as nw
    from oracle import neuconw_oracle as O

    emb, neuconw, nerf, rdr = build_system(seed=2, prec=nw.PREC_F32)
    rays, ts, label, rgbs = synth_rays(1, 5, 64)
    out = rdr.render(rays.cuda(), ts.cuda(), label.cuda(), perturb_overwrite=0,
                     background_rgb=torch.zeros(1, 3).cuda(), cos_anneal_ratio=1.0)
    loss_from_outputs(out, rgbs.cuda()).backward()
    sd = state_dict_cpu(emb, neuconw, nerf, torch.float64)
    ref = O.render(sd, dict(CFG, n_samples=16, n_importance=16), rays.double(), ts, label, 1.0,
                   torch.zeros(1, 3, dtype=torch.float64))
    errs = {k: rel_err(out[k].detach().cpu(), ref[k]) for k in ("color", "depth", "weights_sum")}
    print("single ray, f32:", {k: "%.2e" % v for k, v in errs.items()})
    assert max(errs.values()) < 1e-4, errs
    assert all(torch.isfinite(p.grad).all() for p in neuconw.sdf_net.parameters())


def test_max_samples_per_ray_and_one_past():
    """S + O = 512 is the STANDARD per-ray kernels' LDS capacity (RAY_MAXN): it must work and match the oracle; more goes to the
    large-ray object (test_more_than_512_samples_per_ray); beyond 1088 must raise instead of truncating."""
    import neuralrecon_w_amd as nw
    from oracle import neuconw_oracle as O

    ns, ni = 254, 254  # S = 508, + 4 outside = 512
    emb, neuconw, nerf, rdr = build_system(seed=3, prec=nw.PREC_F32, n_samples=ns, n_importance=ni)
    rays, ts, label, rgbs = synth_rays(6, 9, 64)
    out = rdr.render(rays.cuda(), ts.cuda(), label.cuda(), perturb_overwrite=0,
                     background_rgb=torch.zeros(1, 3).cuda(), cos_anneal_ratio=0.5)
    assert out["weights"].shape == (6, 512)
    loss_from_outputs(out, rgbs.cuda()).backward()
    sd = state_dict_cpu(emb, neuconw, nerf, torch.float64)
    ref = O.render(sd, dict(CFG, n_samples=ns, n_importance=ni), rays.double(), ts, label, 0.5,
                   torch.zeros(1, 3, dtype=torch.float64))
    errs = {k: rel_err(out[k].detach().cpu(), ref[k]) for k in ("color", "depth", "weights_sum")}
    print("508 + 4 samples per ray, f32:", {k: "%.2e" % v for k, v in errs.items()})
    assert max(errs.values()) < 1e-4, errs
    with pytest.raises(ValueError, match="1088"):  # beyond the large-ray kernels' capacity: refused at construction, with the reason
        build_system(seed=3, prec=nw.PREC_F32, n_samples=544, n_importance=544)
    # ... and the C ABI itself refuses an over-long ray instead of truncating it (600 samples: the large-ray object takes it)
    from neuralrecon_w_amd import rayops
    a, b = torch.rand(4, 300, device="cuda"), torch.rand(4, 300, device="cuda")
    merged, _ = rayops.sort_merge(a, b)
    assert torch.equal(merged, torch.sort(torch.cat([a, b], -1), -1)[0])
    with pytest.raises(nw.NeuconwHipError):
        rayops.sort_merge(torch.rand(4, 600, device="cuda"), torch.rand(4, 600, device="cuda"))
    torch.cuda.synchronize()


def test_bad_arguments_fail_loudly():
    import neuralrecon_w_amd as nw
    from neuralrecon_w_amd import voxel

    with pytest.raises((nw.NeuconwHipError, ValueError)):
        voxel.octree_from_points(torch.rand(10, 3, device="cuda"), voxel_size=1e-6, scene_origin=[0, 0, 0], scale=1.0)
    emb, neuconw, nerf, rdr = build_system(seed=4, prec=nw.PREC_F32)
    rays, ts, label, _ = synth_rays(4, 1, 64)
    with pytest.raises(nw.NeuconwHipError):  # host tensors: there is no CPU fallback
        rdr.render(rays, ts, label)


@pytest.mark.parametrize("ns,ni,n_out,steps", [(256, 256, 32, 4), (512, 512, 32, 4)])
def test_more_than_512_samples_per_ray(ns, ni, n_out, steps):
    """config/defaults.py:8-9,31-32 (N_SAMPLES = N_IMPORTANCE = 512, UP_SAMPLE_STEP 4, N_OUTSIDE 32): 1056 samples per ray.  The per-ray
    kernels' large-ray object (csrc/ncw_rays.hip, -DNCW_RAYS_BIG: 1088 samples in LDS) takes over above 512; render + loss + backward
    against the fp32 oracle on a few rays."""
    import neuralrecon_w_amd as nw
    from oracle import neuconw_oracle as O
    from tests._build import build_system, loss_from_outputs, named_params, state_dict_cpu
    from tests._util import rel_err, synth_rays

    emb, neuconw, nerf, rdr = build_system(prec=nw.PREC_F32, n_samples=ns, n_importance=ni, n_outside=n_out, up_sample_steps=steps, seed=2)
    R = 6
    rays, ts, label, rgbs = synth_rays(R, 9, 64)
    out = rdr.render(rays.cuda(), ts.cuda(), label.cuda(), perturb_overwrite=0, background_rgb=torch.zeros(1, 3).cuda(), cos_anneal_ratio=0.5)
    assert out["weights"].shape == (R, ns + ni + n_out)
    loss = loss_from_outputs(out, rgbs.cuda())
    loss.backward()
    sd = {k: v.requires_grad_(True) for k, v in state_dict_cpu(emb, neuconw, nerf, torch.float64).items()}
    cfg = dict(n_samples=ns, n_importance=ni, n_outside=n_out, up_sample_steps=steps, s_val_base=3, render_bg=True, trim_sphere=True,
               mesh_mask_list=["sky"], depth_loss=True, igr_weight=0.1, mask_weight=0.1, depth_weight=0.1, skip_in=(4,), multires=6, multires_view=4)
    ref = O.render(sd, cfg, rays.double(), ts, label, 0.5, torch.zeros(1, 3, dtype=torch.float64))
    lref = O.neuconw_loss(ref, rgbs.double(), cfg)
    g = torch.autograd.grad(lref, sd["neuconw.sdf_net.lin4.weight_v"])[0]
    errs = {k: rel_err(out[k].detach().cpu(), ref[k]) for k in ("color", "depth", "weights_sum", "gradient_error")}
    e_g = rel_err(named_params(emb, neuconw, nerf)["neuconw.sdf_net.lin4.weight_v"].grad.cpu(), g)
    print("%d + %d + %d samples per ray:" % (ns, ni, n_out), {k: "%.2e" % v for k, v in errs.items()}, "d(lin4.weight_v) %.2e" % e_g)
    assert max(errs.values()) < 1e-4 and e_g < 2e-3, (errs, e_g)
    z = out["weights"]
    assert bool(torch.isfinite(z).all()) and bool((z >= 0).all())
