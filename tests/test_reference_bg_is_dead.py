"""CPU, against the REAL reference (runs only where /root/reference is mounted): the premise of the product's
dead-background elimination.  With trim_sphere, NeuconWRenderer.render_core
(/root/reference/rendering/renderer.py:570-783) multiplies the background NeRF's alpha and colour at every primary
sample inside the unit sphere by 1 - inside_sphere = 0 (:637, :693-708): whatever render_core_outside returned there
never reaches an output, and no gradient flows back into it.  Also pins oracle.bg_needed, the CPU statement of the
selection ncw_bg_select makes on the device."""
import os

import pytest
import torch

from tests._util import synth_rays

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/models"), reason="reference tree not mounted")


def _case():
    from oracle import ref_import
    from tests.golden.make_golden import build_reference

    ns = ref_import.load()
    emb, neuconw, nerf, renderer = build_reference(ns, 64, 2, (), seed=2, n_samples=32, n_importance=0)
    R, S, O = 48, 32, 4
    rays, ts, label, rgbs = synth_rays(R, 21, 64)
    o, d = rays[:, 0:3], rays[:, 3:6]
    near, far = torch.full((R, 1), 0.6), torch.full((R, 1), 3.6)  # the interval leaves the unit sphere on both ends
    z = near + (far - near) * torch.linspace(0, 1, S)[None]
    sample_dist = (far - near) / S
    z_out = far / torch.flip(torch.linspace(1e-3, 1 - 1 / (O + 1.0), O), dims=[-1]) + 1.0 / S
    zf, _ = torch.sort(torch.cat([z, z_out], -1), -1)
    a = emb(ts)
    return renderer, nerf, o, d, z, zf, sample_dist, a, rgbs, R, S, O


def test_reference_ignores_background_at_inside_samples():
    renderer, nerf, o, d, z, zf, sample_dist, a, rgbs, R, S, O = _case()
    with torch.no_grad():
        ro = renderer.render_core_outside(o, d, zf, sample_dist, nerf, a_embedded=a)

    def core(bg_alpha, bg_col):
        rc = renderer.render_core(o, d, z, sample_dist, a, cos_anneal_ratio=0.3, background_alpha=bg_alpha,
                                  background_sampled_color=bg_col, background_rgb=torch.zeros(1, 3))
        loss = (rc["color"] - rgbs).abs().sum() / R + 0.1 * rc["gradient_error"] + 0.05 * rc["weights_sum"].mean() \
            + 0.05 * rc["depth"].mean() + 0.01 * rc["color_bg"].sum()
        return rc, loss

    al = ro["alpha"].clone().requires_grad_(True)
    col = ro["sampled_color"].clone().requires_grad_(True)
    rc0, loss0 = core(al * 1.0, col * 1.0)  # (render_core writes into background_alpha in place, :706: no leaves)
    g_al, g_col = torch.autograd.grad(loss0, [al, col])
    ins = rc0["inside_sphere"] > 0  # [R, S]
    assert 0.1 < float(ins.float().mean()) < 0.9
    # no gradient reaches the background values of an inside sample
    assert float(g_al[:, :S][ins].abs().max()) == 0.0 and float(g_col[:, :S][ins].abs().max()) == 0.0
    assert float(g_al[:, :S][~ins].abs().max()) > 0.0 and float(g_al[:, S:].abs().max()) > 0.0
    # and arbitrary values there change nothing the reference returns
    torch.manual_seed(0)
    al2, col2 = ro["alpha"].clone(), ro["sampled_color"].clone()
    al2[:, :S][ins] = torch.rand(int(ins.sum()))
    col2[:, :S][ins] = torch.randn(int(ins.sum()), 3) * 5.0
    rc1, loss1 = core(al2, col2)
    for k in ("color", "depth", "weights", "weights_sum", "color_bg", "gradient_error"):
        assert torch.equal(rc0[k], rc1[k]), k
    assert float(loss0) == float(loss1)


def test_oracle_selection_is_the_reference_mask():
    from oracle import neuconw_oracle as O

    renderer, nerf, o, d, z, zf, sample_dist, a, rgbs, R, S, O_ = _case()
    with torch.no_grad():
        ro = renderer.render_core_outside(o, d, zf, sample_dist, nerf, a_embedded=a)
    rc = renderer.render_core(o, d, z, sample_dist, a, cos_anneal_ratio=0.3, background_alpha=ro["alpha"],
                              background_sampled_color=ro["sampled_color"], background_rgb=torch.zeros(1, 3))
    need = O.bg_needed(o, d, z, sample_dist, O_)
    assert need.shape == (R, S + O_) and bool(need[:, S:].all())
    assert torch.equal(need[:, :S], ~(rc["inside_sphere"] > 0))
