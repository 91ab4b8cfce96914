"""Config 3 (voxel-guided sampling): native ray / voxel near-far against the brute-force slab oracle,
and the whole render() with a fine octree window + boundary samples against the oracle."""
import math

import pytest
import torch

from tests._build import build_system, state_dict_cpu
from tests._util import rel_err, synth_rays

pytestmark = pytest.mark.gpu


def _shell(level, r0=0.5, thick=0.06):
    G = 1 << level
    c = (torch.arange(G).float() + 0.5) * (2.0 / G) - 1.0
    x, y, z = torch.meshgrid(c, c, c, indexing="ij")
    rad = (x * x + y * y + z * z).sqrt()
    return (rad - r0).abs() < thick


@pytest.mark.parametrize("level", [4, 6])
def test_ray_voxel_near_far_vs_bruteforce(level):
    from neuralrecon_w_amd import voxel
    from oracle import neuconw_oracle as O

    occ = _shell(level)
    origin, scale = torch.tensor([0.1, -0.05, 0.2]), 1.7
    R = 500
    rays, _, _, _ = synth_rays(R, 5, 10)
    o = rays[:, 0:3] * scale + origin  # SfM-space origins looking at the shell
    d = rays[:, 3:6]
    o[:50] = origin + 0.01 * torch.randn(50, 3)           # rays starting inside the cube, at the shell centre
    d[50:60] = torch.nn.functional.normalize(torch.randn(10, 3), dim=-1)  # some rays that miss
    o[50:60] = origin + torch.tensor([5.0, 5.0, 5.0])
    near_ref, far_ref = O.ray_voxel_near_far(o.double(), d.double(), occ, origin.double(), scale)
    od = voxel.occupancy_from_dense(occ.cuda(), origin, scale)
    near, far = voxel.get_near_far(o.cuda(), d.cuda(), od)
    near, far = near.cpu(), far.cpu()
    hit_ref = near_ref > 0
    assert torch.equal(near > 0, hit_ref)
    assert int(hit_ref.sum()) > 300 and int((~hit_ref).sum()) >= 10
    assert rel_err(near[hit_ref], near_ref[hit_ref]) < 1e-5
    assert rel_err(far[hit_ref], far_ref[hit_ref]) < 1e-5
    assert bool((far >= near).all())


def _sparse_shell(level, n_keep, seed, r0=None):
    """Occupied voxels of a one-voxel-thick sphere shell at a level too fine for a dense tensor (10: 2^30 voxels) as an index list
    [V,3]: a COMPLETE shell of about n_keep voxels (radius chosen for that, <= 0.5) + 5 % clutter.  Built from surface points, not
    from the grid."""
    G = 1 << level
    g = torch.Generator().manual_seed(seed)
    if r0 is None:  # 4 pi (r G / 2)^2 ~ n_keep
        r0 = min(0.5, math.sqrt(n_keep / (4 * math.pi)) * 2.0 / G)
    p = torch.nn.functional.normalize(torch.randn(40 * n_keep, 3, generator=g, dtype=torch.float64), dim=-1) * r0
    q = torch.unique(torch.floor((p + 1.0) * (G / 2)).long().clamp(0, G - 1), dim=0)
    clutter = torch.randint(0, G, (n_keep // 20, 3), generator=g)
    return torch.unique(torch.cat([q, clutter]), dim=0), G, r0


def _rays_at_shell(R, origin, scale, seed, r0):
    """Rays from outside the cube aimed INTO the sphere (they all cross the shell), 20 from inside it, 10 that miss the cube."""
    g = torch.Generator().manual_seed(seed)
    o = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1) * 2.5
    target = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1) * (0.8 * r0 * torch.rand(R, 1, generator=g))
    d = torch.nn.functional.normalize(target - o, dim=-1)
    o[:20] = 0.3 * r0 * torch.randn(20, 3, generator=g)
    d[20:30] = torch.nn.functional.normalize(torch.randn(10, 3, generator=g), dim=-1)
    o[20:30] = torch.tensor([5.0, 5.0, 5.0])
    return o * scale + origin, d


@pytest.mark.parametrize("level,n_keep", [(8, 20000), (10, 30000)])
def test_ray_voxel_near_far_at_the_reference_levels(level, n_keep):
    """Octree level 10 is what the reference runs at (scripts/sdf_extract.sh:13-17, README.md:61): 128 MB bit grid, 16 M bricks.
    The fp64 slab oracle tests the rays against the OCCUPIED voxels only.  A float32 DDA and an fp64 slab test may disagree on a
    voxel the ray merely grazes, so the comparison is a sandwich: counting grazing contacts (tmax - tmin > -eps) can only move
    near down / far up, discounting them (> +eps) the other way; the kernel must lie between the two, and equal the exact test on
    all but a handful of rays."""
    from neuralrecon_w_amd import voxel
    from oracle import neuconw_oracle as O

    idx, G, r0 = _sparse_shell(level, n_keep, level)
    origin, scale = torch.tensor([0.1, -0.05, 0.2]), 1.7
    R = 384
    o, d = _rays_at_shell(R, origin, scale, 5, r0)
    centres = (idx.float() + 0.5) * (2.0 / G) - 1.0
    od = voxel.occupancy_from_points((centres * scale + origin).cuda(), origin, scale, level)
    assert torch.equal(voxel.voxels_from_occupancy(od).cpu(), idx)  # the bit grid holds exactly the listed voxels
    near, far = [t.cpu().double() for t in voxel.get_near_far(o.cuda(), d.cuda(), od)]
    eps = 2e-3 * (2.0 / G)  # a crossing shorter than 0.2 % of a voxel is "grazing": float32 grid coordinates resolve ~1e-4 voxel at level 10
    ex = O.ray_voxel_near_far(o.double(), d.double(), (idx, G), origin.double(), scale)
    lo = O.ray_voxel_near_far(o.double(), d.double(), (idx, G), origin.double(), scale, margin=-eps)  # with grazing contacts
    hi = O.ray_voxel_near_far(o.double(), d.double(), (idx, G), origin.double(), scale, margin=eps)   # without
    hit = ex[0] > 0
    assert int(hit.sum()) > 200 and int((~hit).sum()) >= 10
    tol = 1e-5 * scale * 5.0  # float32 depths up to ~5 cube units (the existing level-4 / 6 test: 1e-5 of the largest depth)
    sure = (lo[0] > 0) == (hi[0] > 0)  # hit / miss does not hinge on a grazing contact
    assert torch.equal((near > 0)[sure], hit[sure])
    both = (near > 0) & (lo[0] > 0) & (hi[0] > 0)
    assert bool((near[both] >= lo[0][both] - tol).all()) and bool((near[both] <= hi[0][both] + tol).all())
    assert bool((far[both] <= lo[1][both] + tol).all()) and bool((far[both] >= hi[1][both] - tol).all())
    same = hit & (near > 0) & ((near - ex[0]).abs() < tol) & ((far - ex[1]).abs() < tol)
    print("level %d: %d voxels, %d / %d rays hit, %d identical to the exact test, %d decided by a grazing contact" %
          (level, idx.shape[0], int(hit.sum()), R, int(same.sum()), int((~sure).sum())))
    assert int(same.sum()) >= int(hit.sum()) - max(3, R // 100)
    assert bool((far >= near).all())


@pytest.mark.parametrize("level,n_keep,with_exit", [(5, 400, True), (8, 20000, False), (10, 30000, True)])
def test_kaolin_unbatched_raytrace_vs_bruteforce(level, n_keep, with_exit):
    """compat/kaolin `render.spc.unbatched_raytrace` (generate_voxel.py:358-368): ALL nuggets, ordered by ray then depth, point
    ids into the point hierarchy, depths (entry[, exit]) -- against the fp64 brute-force list; and the first / last nugget per
    ray (what get_near_far keeps, :376-395) against ncw_ray_voxel_near_far's fused answer."""
    from oracle import neuconw_oracle as O
    from tests._util import compat_kaolin

    spc, spc_render = compat_kaolin()
    from neuralrecon_w_amd import voxel

    idx, G, r0 = _sparse_shell(level, n_keep, 100 + level)
    R = 256
    o, d = _rays_at_shell(R, torch.zeros(3), 1.0, 9, r0)
    octree = spc.unbatched_points_to_octree(idx.short().cuda(), level)
    _, pyramid, prefix = spc.scan_octrees(octree, torch.tensor([len(octree)], dtype=torch.int32))
    points = spc.generate_points(octree, pyramid, prefix)
    pyramid = pyramid[0]
    leaf0 = int(pyramid[1, level])
    assert int(pyramid[0, level]) == idx.shape[0]
    ray, pid, depth = spc_render.unbatched_raytrace(octree, points, pyramid, prefix, o.cuda(), d.cuda(), level, return_depth=True,
                                                    with_exit=with_exit)
    assert depth.shape == (ray.shape[0], 2 if with_exit else 1) and pid.shape == ray.shape
    ray, pid, depth = ray.cpu().long(), pid.cpu().long(), depth.cpu().double()
    vox = points.cpu()[pid].long()  # the nuggets' voxels
    assert bool((pid >= leaf0).all())
    # ordered by ray, then by entry depth
    assert bool((ray[1:] >= ray[:-1]).all())
    same_ray = ray[1:] == ray[:-1]
    assert bool((depth[1:, 0][same_ray] >= depth[:-1, 0][same_ray]).all())
    if with_exit:
        assert bool((depth[:, 1] >= depth[:, 0]).all())
    # brute force: the nugget SET between the two grazing brackets, depths of the common nuggets
    eps = 2e-3 * (2.0 / G)  # a crossing shorter than 0.2 % of a voxel is "grazing": float32 grid coordinates resolve ~1e-4 voxel at level 10
    key = lambda r, v: r * (G ** 3) + (v[:, 0] * G + v[:, 1]) * G + v[:, 2]  # noqa: E731
    got = key(ray, vox)
    r_lo, v_lo, dep_lo = O.ray_voxel_nuggets(o.double(), d.double(), idx, G, margin=-eps)
    r_hi, v_hi, _ = O.ray_voxel_nuggets(o.double(), d.double(), idx, G, margin=eps)
    k_lo, k_hi = key(r_lo, idx[v_lo]), key(r_hi, idx[v_hi])
    assert got.unique().shape[0] == got.shape[0]
    assert bool(torch.isin(got, k_lo).all()), "a reported nugget is not a crossing"
    assert bool(torch.isin(k_hi, got).all()), "a clear crossing is missing"
    srt = torch.argsort(k_lo)
    at = srt[torch.searchsorted(k_lo[srt], got)]
    tol = 5e-5  # float32 depths up to ~5
    assert float((depth[:, 0] - dep_lo[at, 0]).abs().max()) < tol
    if with_exit:
        assert float((depth[:, 1] - dep_lo[at, 1]).abs().max()) < tol
    print("raytrace level %d: %d nuggets on %d rays (brackets %d .. %d)" % (level, got.shape[0], int(ray.unique().shape[0]), k_hi.shape[0], k_lo.shape[0]))
    assert got.shape[0] > R
    # first / last nugget per ray == the fused near / far kernel (same walk; it adds get_near_far's 1e-7 offsets itself)
    od = voxel.OctreeData({"octree": octree, "spc_data": {"points": points, "pyramid": pyramid, "prefix": prefix}, "level": level,
                           "scale": 1.0, "scene_origin": torch.zeros(3)})
    near, far = [t.cpu().double().reshape(-1) for t in voxel.get_near_far(o.cuda(), d.cuda(), od)]
    first = torch.ones(ray.shape[0], dtype=torch.bool)
    first[1:] = ray[1:] != ray[:-1]
    last = torch.ones(ray.shape[0], dtype=torch.bool)
    last[:-1] = ray[1:] != ray[:-1]
    n_tr, f_tr = torch.zeros(R, dtype=torch.float64), torch.zeros(R, dtype=torch.float64)
    n_tr[ray[first]], f_tr[ray[last]] = depth[first, 0], depth[last, 0]
    ok = n_tr > 1e-4  # generate_voxel.py:397
    agree = ((near - torch.where(ok, n_tr, 0 * n_tr)).abs() < 1e-5) & ((far - torch.where(ok, f_tr, 0 * f_tr)).abs() < 1e-5)
    assert int((~agree).sum()) <= 5, int((~agree).sum())  # (the fused kernel's own 1e-7 offsets decide a grazing contact: measured 0 .. 3 of 256)


# (W, precision, rays, tol per-ray outputs, tol per-sample tensors).  f32: the W = 64 networks of round 1; f16: the HEADLINE
# networks (W = 256: the split-precision SDF value path exists there) in the timed dtype, at the fp16 output tolerance of
# networks (W = 256: the split-precision SDF value path exists there) in the timed dtype -- config 3 in the precision `bench.py --config
# voxel` times.  Every per-ray output bound is the north-star bar, 1e-4 (measured: fp32 2.3e-6 / 1.0e-5, fp16 4.3e-5, mask_error 6.7e-5).
@pytest.mark.parametrize("W,prec_name,R,tol_out,tol_sample", [(64, "f32", 96, 1e-4, 2e-3), (256, "f32", 48, 1e-4, 2e-3),
                                                              (256, "f16", 48, 1e-4, 2e-3)])
def test_render_with_fine_octree_window_and_boundary_samples(W, prec_name, R, tol_out, tol_sample):
    """sampler narrowed to surface +- SAMPLE_RANGE voxels, 10 boundary samples (renderer.py:415-456, 549-566)."""
    import neuralrecon_w_amd as nw
    from neuralrecon_w_amd import voxel
    from oracle import neuconw_oracle as O

    level = 6
    occ = _shell(level, 0.5, 0.05)
    prec = {"f32": nw.PREC_F32, "f16": nw.PREC_F16}[prec_name]
    big = dict(n_a=48, n_vocab=100, nerf_w=256, color_hidden=256, head=128) if W == 256 else {}
    emb, neuconw, nerf, rdr = build_system(W=W, prec=prec, n_samples=16, n_importance=16, boundary_samples=10,
                                           sample_range=4, seed=11, **big)
    with torch.no_grad():
        for n, p in neuconw.named_parameters():
            if n.endswith("weight_g"):
                p.mul_(1.0 + 0.05 * torch.randn_like(p))
    vs = 2.0 / (1 << level)
    rdr.fine_octree_data = voxel.occupancy_from_dense(occ.cuda(), torch.zeros(3), 1.0, voxel_size=vs)
    rays, ts, label, rgbs = synth_rays(R, 8, 64)
    out = rdr.render(rays.cuda(), ts.cuda(), label.cuda(), perturb_overwrite=0, background_rgb=torch.zeros(1, 3).cuda(),
                     cos_anneal_ratio=0.5)
    odt = torch.float32 if prec_name == "f32" and W == 64 else torch.float64  # fp64 arbitrates at the headline width
    sd = state_dict_cpu(emb, neuconw, nerf, odt)
    cfg = dict(n_samples=16, n_importance=16, n_outside=4, up_sample_steps=2, s_val_base=3, render_bg=True,
               trim_sphere=True, mesh_mask_list=["sky"], depth_loss=True, skip_in=(4,), multires=6, multires_view=4,
               boundary_samples=10, sample_range=4, radius=1.0)
    fine = dict(occ=occ, scene_origin=torch.zeros(3), scale=1.0, voxel_size=vs)
    ref = O.render(sd, cfg, rays.to(odt), ts, label, 0.5, torch.zeros(1, 3, dtype=odt), fine_octree=fine)
    assert out["weights"].shape == ref["weights"].shape == (R, 16 + 16 + 10 + 4)
    errs = {k: rel_err(out[k].detach().cpu(), ref[k]) for k in ("color", "depth", "weights_sum", "gradient_error", "mask_error",
                                                                "sfm_depth_loss", "color_bg", "weights", "cdf_fine", "gradients")}
    print("voxel-guided render W=%d %s:" % (W, prec_name), {k: "%.2e" % v for k, v in errs.items()})
    for k in ("color", "depth", "weights_sum", "mask_error"):
        assert errs[k] < tol_out, (k, errs[k])
    # fp16: sfm_depth_loss = w (depth - depth_gt)^2 amplifies the depth error by 2 depth / |depth - depth_gt| (measured 1.0e-4); color_bg is the
    # background NeRF's share alone, relative to ITS small maximum -- 2.8e-4 with plain fp16 operands, 1.0e-6 since the split-precision
    # refinement of the usable samples (ncw_nerf_refine): the common bound
    assert errs["sfm_depth_loss"] < (2.5e-4 if prec_name == "f16" else tol_out), errs["sfm_depth_loss"]
    assert errs["color_bg"] < tol_out, errs["color_bg"]
    # the eikonal term: the adjoint sweep's t_l stays single-rounded fp16 (per-sample normals 2.8e-4; measured 1.2e-4)
    assert errs["gradient_error"] < (3e-4 if prec_name == "f16" else tol_out), errs["gradient_error"]
    for k in ("weights", "cdf_fine", "gradients"):
        assert errs[k] < tol_sample, (k, errs[k])
