"""GPU parity of the per-ray sampler / compositor kernels (C ABI) against the CPU oracle and the
golden vectors of the real reference."""
import pytest
import torch

from tests._util import load_golden, rel_err, synth_rays

pytestmark = pytest.mark.gpu


def _cu(*ts):
    return [t.cuda() if t is not None else None for t in ts]


def test_sample_pdf_upsample_golden():
    """up_sample (+sample_pdf) and cat_z_vals against vectors produced by the reference itself."""
    from neuralrecon_w_amd import rayops

    _, _, _, m = load_golden("units_w64")
    r = m["us_rays"]
    znew = rayops.upsample(*_cu(r[:, 0:3], r[:, 3:6], m["us_z"], m["us_sdf"]), 8, 64 * 2 ** 3).cpu()
    assert rel_err(znew, m["us_znew"]) < 1e-5
    # cat_z_vals: merge with payload. sdf_new is recovered from the golden merged array
    zc, _ = rayops.sort_merge(*_cu(m["us_z"], m["us_znew"]))
    assert torch.equal(zc.cpu(), m["us_zcat"])


@pytest.mark.parametrize("n,n_new,inv_s", [(64, 32, 512.0), (96, 32, 1024.0), (8, 8, 512.0), (17, 5, 64.0)])
def test_upsample_vs_oracle(n, n_new, inv_s):
    from neuralrecon_w_amd import rayops
    from oracle import neuconw_oracle as O

    R = 301
    rays, _, _, _ = synth_rays(R, 3, 10)
    o, d = rays[:, 0:3], rays[:, 3:6]
    g = torch.Generator().manual_seed(n)
    z = torch.sort(1.0 + 2.0 * torch.rand(R, n, generator=g), -1)[0]
    pts = o[:, None] + d[:, None] * z[..., None]
    sdf = pts.norm(dim=-1) - 0.5 + 0.02 * torch.randn(R, n, generator=g)
    ref = O.up_sample(o, d, z, sdf, n_new, inv_s)
    got = rayops.upsample(*_cu(o, d, z, sdf), n_new, inv_s).cpu()
    assert rel_err(got, ref) < 2e-5
    assert bool((got[:, 1:] >= got[:, :-1]).all())  # inverse-CDF samples are sorted


def test_sort_merge_properties():
    from neuralrecon_w_amd import rayops

    g = torch.Generator().manual_seed(0)
    a = torch.sort(torch.rand(257, 70, generator=g), -1)[0]
    b = torch.rand(257, 33, generator=g)  # unsorted second operand (boundary samples case)
    b[:, 0] = a[:, 5]  # exact ties
    pa, pb = torch.rand(257, 70, generator=g), torch.rand(257, 33, generator=g)
    out, pout = rayops.sort_merge(*_cu(a, b, pa, pb))
    ref, idx = torch.sort(torch.cat([a, b], -1), dim=-1, stable=True)
    assert torch.equal(out.cpu(), ref)
    assert torch.equal(pout.cpu(), torch.gather(torch.cat([pa, pb], -1), 1, idx))


@pytest.mark.parametrize("perturb", [False, True])
def test_sample_coarse_vs_oracle(perturb):
    from neuralrecon_w_amd import rayops
    from oracle import neuconw_oracle as O

    R, n, no = 130, 64, 4
    rays, _, _, _ = synth_rays(R, 4, 10)
    near, far = rays[:, 6:7], rays[:, 7:8] + 0.3 * torch.rand(R, 1)
    rs = torch.rand(R, 1) if perturb else None
    ro = torch.rand(R, no) if perturb else None
    cfg = dict(n_samples=n, n_importance=0, n_outside=no, up_sample_steps=1, s_val_base=0, render_bg=True)
    z_ref, zo_ref, sd_ref = O.sparse_sampler({}, cfg, rays[:, 0:3], rays[:, 3:6], near, far, rs, ro)
    z, zo, sd = rayops.sample_coarse(*_cu(near, far, near, far), n, no, *(_cu(rs, ro)))
    assert rel_err(z.cpu(), z_ref) < 1e-6
    assert rel_err(zo.cpu(), zo_ref) < 1e-6
    assert rel_err(sd.cpu(), sd_ref) < 1e-6


def _comp_inputs(R, S, O_, seed, with_bg=True):
    g = torch.Generator().manual_seed(seed)
    rays, _, _, _ = synth_rays(R, seed, 10)
    o, d = rays[:, 0:3], rays[:, 3:6]
    z = torch.sort(1.0 + 2.2 * torch.rand(R, S, generator=g), -1)[0]
    sample_dist = torch.full((R, 1), 2.0 / S)
    mid = z + torch.cat([z[:, 1:] - z[:, :-1], sample_dist], -1) * 0.5
    pts = o[:, None] + d[:, None] * mid[..., None]
    sdf = pts.norm(dim=-1) - 0.5 + 0.01 * torch.randn(R, S, generator=g)
    grad = pts / pts.norm(dim=-1, keepdim=True) * (1 + 0.1 * torch.randn(R, S, 1, generator=g)) \
        + 0.05 * torch.randn(R, S, 3, generator=g)
    rgb = torch.rand(R, S, 3, generator=g)
    z_out = 3.3 + torch.sort(torch.rand(R, O_, generator=g), -1)[0] * 5
    z_feed = torch.sort(torch.cat([z, z_out], -1), -1)[0] if with_bg else None
    density = torch.randn(R, S + O_, generator=g) * 2 if with_bg else None
    bg_rgb = torch.rand(R, S + O_, 3, generator=g) if with_bg else None
    inv_s = torch.tensor([20.0])
    return dict(o=o, d=d, z=z, sample_dist=sample_dist, sdf=sdf, grad=grad, rgb=rgb, z_feed=z_feed, density=density,
                bg_rgb=bg_rgb, inv_s=inv_s)


@pytest.mark.parametrize("S,O_,with_bg,brgb", [(128, 4, True, True), (24, 4, True, False), (32, 0, False, True),
                                               (130, 32, True, False)])
def test_composite_fwd_bwd_vs_oracle(S, O_, with_bg, brgb):
    from neuralrecon_w_amd import rayops
    from oracle import neuconw_oracle as O

    R = 77
    I = _comp_inputs(R, S, O_, 7 + S, with_bg)
    cosr = 0.3
    background_rgb = torch.tensor([[0.1, 0.2, 0.3]]) if brgb else None
    # ---- oracle (fp64 arbitrates the fp32 kernels) ------------------------------------------
    leaf = {k: I[k].double().requires_grad_(True) for k in ("sdf", "grad", "rgb", "inv_s")}
    if with_bg:
        leaf["density"] = I["density"].double().requires_grad_(True)
        leaf["bg_rgb"] = I["bg_rgb"].double().requires_grad_(True)
        bg_alpha = O.bg_alpha_from_density(leaf["density"], I["z_feed"].double(), I["sample_dist"].double())
    ref = O.composite(dict(trim_sphere=True), I["o"].double(), I["d"].double(), I["z"].double(),
                      I["sample_dist"].double(), leaf["rgb"], leaf["inv_s"], leaf["sdf"], leaf["grad"], cosr,
                      bg_alpha if with_bg else None, leaf["bg_rgb"] if with_bg else None,
                      background_rgb.double() if brgb else None)
    # ---- HIP -----------------------------------------------------------------------------------
    ctx = rayops.CompositeCtx(*_cu(I["o"], I["d"], I["z"], I["sample_dist"], I["sdf"], I["grad"], I["rgb"],
                                   I["inv_s"]), cosr, *(_cu(I["z_feed"], I["density"], I["bg_rgb"])),
                              background_rgb=background_rgb.cuda() if brgb else None)
    out = ctx.forward()
    pairs = [("color", "color"), ("color_sphere", "color_sphere"), ("weights", "weights"),
             ("weights_sum", "weights_sum"), ("cdf", "cdf"), ("inside", "inside_sphere"), ("depth", "depth"),
             ("normals", "normals"), ("mid_z", "mid_z_vals"), ("dists", "dists")]
    if with_bg:
        pairs.append(("color_bg", "color_bg"))
    for k, kr in pairs:
        e = rel_err(out[k].cpu().reshape(ref[kr].shape), ref[kr])
        assert e < 2e-5, (k, e)
    assert rel_err(out["eik"][0].cpu(), ref["eik_num"]) < 2e-5
    assert rel_err(out["eik"][1].cpu(), ref["eik_den"]) < 1e-6
    # ---- backward ------------------------------------------------------------------------------
    g = torch.Generator().manual_seed(1)
    dc, dw, dd, de = (torch.randn(R, 3, generator=g), torch.randn(R, generator=g), torch.randn(R, generator=g),
                      torch.randn(R, generator=g))
    loss = (ref["color"] * dc.double()).sum() + (ref["weights_sum"][:, 0] * dw.double()).sum() \
        + (ref["depth"] * dd.double()).sum() + (ref["eik_num"] * de.double()).sum()
    names = list(leaf)
    gref = dict(zip(names, torch.autograd.grad(loss, [leaf[k] for k in names])))
    got = ctx.backward(*_cu(dc, dw, dd, de))
    m = {"sdf": "d_sdf", "grad": "d_grad", "rgb": "d_rgb", "inv_s": "d_inv_s", "density": "d_density",
         "bg_rgb": "d_bg_rgb"}
    got["d_inv_s"] = got["d_inv_s"].sum().reshape(1)  # per-ray terms: the caller reduces them (ncw_inv_s_bwd in the renderer)
    for k in names:
        e = rel_err(got[m[k]].cpu().reshape(gref[k].shape), gref[k])
        assert e < 1e-4, (k, e)
