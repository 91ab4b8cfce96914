"""Live pinning of the oracle against the REAL reference at the full network widths (W=256 and the
shipped W=512).  Runs only where /root/reference is mounted (the build container)."""
import os

import pytest
import torch

from tests._util import rel_err

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/models"), reason="reference tree not mounted")


@pytest.mark.parametrize("W", [256, 512])
def test_sdf_color_nerf_full_width(W):
    from oracle import neuconw_oracle as O
    from oracle import ref_import

    ns = ref_import.load()
    torch.manual_seed(W)
    sdf_cfg = dict(d_in=3, d_out=W + 1, d_hidden=W, n_layers=8, skip_in=(4,), multires=6, bias=0.5, scale=1,
                   geometric_init=True, weight_norm=True, inside_outside=False)
    color_cfg = dict(d_in=9, d_feature=W, mode="idr", d_out=3, d_hidden=256, n_layers=4, head_channels=128,
                     static_head_layers=2, weight_norm=True, multires_view=4)
    net = ns.NeuconW(sdf_cfg, color_cfg, dict(init_val=0.3), 48, True)
    bg = ns.NeRF(D=8, d_in=4, d_in_view=3, W=256, multires=10, multires_view=4, output_ch=4, skips=[4],
                 encode_appearance=True, in_channels_a=48, in_channels_dir=27, use_viewdirs=True)
    with torch.no_grad():
        for n, p in net.named_parameters():
            if n.endswith("weight_g"):
                p.mul_(1.0 + 0.1 * torch.randn_like(p))
    R, S = 6, 8
    x = torch.cat([(torch.rand(R, S, 3) * 2 - 1) * 0.8, torch.nn.functional.normalize(torch.randn(R, S, 3), dim=-1),
                   torch.randn(R, S, 48) * 0.3], -1)
    rgb, inv_s, sdf, grad = net(x)
    loss = (rgb ** 2).sum() + sdf.sum() + (grad ** 2).sum()
    names = [n for n, p in net.named_parameters() if not n.startswith("xyz_encoding_final")]
    gref = torch.autograd.grad(loss, [dict(net.named_parameters())[n] for n in names], allow_unused=True)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    xf = x.reshape(-1, 54)
    rgb_o, inv_s_o, sdf_o, grad_o = O.neuconw_forward(sd, xf[:, 0:3], xf[:, 3:6], xf[:, 6:], dict(skip_in=(4,)))
    assert rel_err(rgb_o, rgb.reshape(-1, 3)) < 2e-5
    assert rel_err(sdf_o, sdf.reshape(-1)) < 2e-5
    assert rel_err(grad_o, grad.reshape(-1, 3)) < 2e-5
    assert rel_err(inv_s_o, inv_s) < 1e-6
    loss_o = (rgb_o ** 2).sum() + sdf_o.sum() + (grad_o ** 2).sum()
    gor = torch.autograd.grad(loss_o, [sd[n] for n in names], allow_unused=True)
    for n, a, b in zip(names, gor, gref):
        assert (a is None) == (b is None), n
        if a is not None:
            assert rel_err(a, b) < 5e-4, (n, rel_err(a, b))
    p4 = torch.cat([torch.nn.functional.normalize(torch.randn(40, 3), dim=-1), torch.rand(40, 1)], -1)
    dens, col = bg(p4, xf[:40, 3:6], xf[:40, 6:])
    dens_o, col_o = O.nerf_net({k: v.detach() for k, v in bg.state_dict().items()}, p4, xf[:40, 3:6], xf[:40, 6:])
    assert rel_err(dens_o, dens) < 2e-5 and rel_err(col_o, col) < 2e-5
