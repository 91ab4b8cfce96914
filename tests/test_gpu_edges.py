"""Edge cases of the hot path through the C ABI: empty inputs, a single ray, the maximum samples-per-ray the
per-ray kernels hold in LDS (S + O = 512) and one past it (must fail loudly), bad arguments."""
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
    for k in ("color", "depth", "weights_sum"):
        assert rel_err(out[k].detach().cpu(), ref[k]) < 2e-4, k
    assert all(torch.isfinite(p.grad).all() for p in neuconw.sdf_net.parameters())


def test_max_samples_per_ray_and_one_past():
    """S + O = 512 is the per-ray kernels' LDS capacity (RAY_MAXN): it must work and match the oracle; 516 must
    raise instead of truncating."""
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
    for k in ("color", "depth", "weights_sum"):
        assert rel_err(out[k].detach().cpu(), ref[k]) < 5e-4, (k, rel_err(out[k].detach().cpu(), ref[k]))
    with pytest.raises(ValueError, match="512"):  # refused at construction, with the reason (config/defaults.py's 512 + 512)
        build_system(seed=3, prec=nw.PREC_F32, n_samples=256, n_importance=256)
    # ... and the C ABI itself refuses an over-long ray instead of truncating it
    from neuralrecon_w_amd import rayops
    with pytest.raises(nw.NeuconwHipError):
        rayops.sort_merge(torch.rand(4, 300, device="cuda"), torch.rand(4, 300, device="cuda"))
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
