"""GPU: bisect the one-in-16k glitch of the split value-only kernels in t-units (round 6): depth sweep + coordinate sweep around the
point the glitch was seen at (x = [0.36032823, 0.55992794, -0.00175765])."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import neuralrecon_w_amd as nw  # noqa: E402
from neuralrecon_w_amd.neuconw import points_struct  # noqa: E402
from neuralrecon_w_amd.stash import StashCache  # noqa: E402
from oracle import neuconw_oracle as O  # noqa: E402
from tests._parity import perturb_weights  # noqa: E402


def build(W, nl, skip):
    torch.manual_seed(0)
    net = nw.SDFNetwork(d_in=3, d_out=W + 1, d_hidden=W, n_layers=nl, skip_in=skip, multires=6, bias=0.5, scale=1, geometric_init=True,
                        weight_norm=True, inside_outside=False)

    class Holder(torch.nn.Module):
        def __init__(self, n):
            super().__init__()
            self.sdf_net = n

    perturb_weights(Holder(net), 0.1, 0.02)
    return net.cuda()


p0 = torch.tensor([0.36032822728157043, 0.5599279403686523, -0.0017576501704752445])
g = torch.Generator().manual_seed(3)
sweeps = {
    "the point + 4095 random points": torch.cat([p0[None], (torch.rand(4095, 3, generator=g) * 2 - 1) * 0.7]),
    "x, y fixed, z swept over +-0.01": torch.stack([p0[0].expand(4096), p0[1].expand(4096), torch.linspace(-0.01, 0.01, 4096)], 1),
    "random x, y; |z| < 0.01": torch.cat([(torch.rand(4096, 2, generator=g) * 2 - 1) * 0.7, (torch.rand(4096, 1, generator=g) * 2 - 1) * 0.01], 1),
    "random points with ONE coordinate < 1e-3": torch.cat([(torch.rand(4096, 2, generator=g) * 2 - 1) * 0.7, (torch.rand(4096, 1, generator=g) * 2 - 1) * 1e-3], 1)[:, [2, 0, 1]],
}
for W in (256,):
    for nl, skip in ((8, (4,)), (8, ()), (4, ()), (2, ()), (1, ())):
        net = build(W, nl, skip)
        sd = {"sdf_net." + k: v.detach().cpu().double() for k, v in net.state_dict().items()}
        net.sdf_split = True
        for name, x in sweeps.items():
            ref = O.sdf_net(sd, x.double(), "sdf_net.", skip_in=skip, with_grad=False)[0]
            s_inf = net.sdf(x.cuda(), nw.PREC_F16).reshape(-1).cpu().double()
            s_fwd, _, c = net.fwd_stash(points_struct(x=x.cuda()), x.shape[0], nw.PREC_F16)
            StashCache.release(c["lease"])
            e_inf, e_fwd = (s_inf - ref).abs(), (s_fwd.cpu().double() - ref).abs()
            bad = (e_inf > 1e-5).nonzero().reshape(-1)
            print("W=%d layers=%d skip=%s | %-44s infer max %.2e (fwd %.2e), %d points above 1e-5%s" % (
                W, nl, skip, name, float(e_inf.max()), float(e_fwd.max()), bad.numel(),
                "" if bad.numel() == 0 else ": first " + ", ".join("%s err %.1e" % (["%.6f" % v for v in x[i].tolist()], float(e_inf[i])) for i in bad[:4].tolist())),
                flush=True)
