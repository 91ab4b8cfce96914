"""GPU: the fused per-ray loss terms of render() (ncw_ray_tail_fwd/bwd through renderer._RayTailFn) against the
torch formulas of the reference (renderer.py:763-765, 869-877, 892-897), values and gradients, incl. the edge
cases: no ray selected for the depth term, masking off, weights_sum outside the BCE clip range."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref(wsum, depth, num, den, label, gt, w, ids, has_mask, has_depth):
    ge = num.sum() / (den.sum() + 1e-5)
    me = sfm = None
    if has_mask:
        mask = torch.ones_like(wsum)
        for i in ids:
            mask[label == i] = 0
        me = F.binary_cross_entropy(wsum.clip(1e-3, 1.0 - 1e-3), mask, reduction="none")
    if has_depth:
        sel = (w > 0).to(depth.dtype)
        sfm = ((depth - gt) ** 2) * w * sel * (float(depth.shape[0]) / sel.sum().clamp_min(1.0))
    return me, sfm, ge


@pytest.mark.parametrize("R,frac_sel,has_mask,has_depth", [(1000, 0.2, True, True), (37, 0.0, True, True),
                                                            (300, 0.5, False, True), (300, 0.5, True, False), (1, 1.0, True, True)])
def test_ray_tail_matches_torch(R, frac_sel, has_mask, has_depth):
    from neuralrecon_w_amd.renderer import _RayTailFn

    g = torch.Generator().manual_seed(R)
    wsum = (torch.rand(R, generator=g) * 1.2 - 0.1).cuda()  # some outside [1e-3, 1-1e-3]: clip kills their gradient
    depth = (torch.rand(R, generator=g) * 3).cuda()
    num, den = torch.rand(R, generator=g).cuda(), (torch.rand(R, generator=g) > 0.3).float().cuda()
    label = torch.randint(0, 4, (R,), generator=g).cuda()
    gt = (torch.rand(R, generator=g) * 3).cuda()
    w = (torch.rand(R, generator=g) * (torch.rand(R, generator=g) < frac_sel)).cuda()
    ids = (2, 3)
    cot = [torch.randn(R, generator=g).cuda(), torch.randn(R, generator=g).cuda(), torch.randn(1, generator=g).cuda()]
    outs = []
    for fused in (False, True):
        a, b, c = (t.clone().requires_grad_(True) for t in (wsum, depth, num))
        if fused:
            me, sfm, ge = _RayTailFn.apply(a, b, c, den, label, gt, w, ids, has_mask, has_depth)
        else:
            me, sfm, ge = _ref(a, b, c, den, label, gt, w, ids, has_mask, has_depth)
        loss = (ge.reshape(1) * cot[2]).sum()
        if has_mask:
            loss = loss + (me * cot[0]).sum()
        if has_depth:
            loss = loss + (sfm * cot[1]).sum()
        loss.backward()
        zero = torch.zeros(R, device="cuda")
        outs.append((me, sfm, ge.reshape(1), a.grad if a.grad is not None else zero, b.grad if b.grad is not None else zero,
                     c.grad))
    for x, y in zip(*outs):
        if x is None or y is None:
            assert x is None and y is None
            continue
        assert torch.allclose(x.detach(), y.detach(), rtol=2e-5, atol=1e-6), float((x - y).abs().max())
