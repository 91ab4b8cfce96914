"""GPU: accuracy and cost of the split-precision SDF value path (csrc/ncw_split.hip) against the plain fp16 kernels, the
exact-fp32 mode and the fp64 oracle; W = 256, 131,072 points."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import neuralrecon_w_amd as nw  # noqa: E402
from neuralrecon_w_amd.neuconw import points_struct  # noqa: E402
from neuralrecon_w_amd.stash import StashCache  # noqa: E402
from oracle import neuconw_oracle as O  # noqa: E402
from tests._parity import perturb_weights  # noqa: E402

torch.manual_seed(0)
net = nw.SDFNetwork(d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=(4,), multires=6, bias=0.5, scale=1,
                    geometric_init=True, weight_norm=True, inside_outside=False)


class Holder(torch.nn.Module):
    def __init__(self, n):
        super().__init__()
        self.sdf_net = n


hold = Holder(net)
perturb_weights(hold, 0.1, 0.0)
net = net.cuda()
N = 131072
g = torch.Generator().manual_seed(1)
x = torch.randn(N, 3, generator=g)
x = (x / x.norm(dim=-1, keepdim=True) * torch.rand(N, 1, generator=g) ** (1 / 3)).float()
xc = x.cuda()
sd = {"sdf_net." + k: v.detach().cpu().double() for k, v in net.state_dict().items()}
M = 8192
ref_sdf, _, ref_grad = O.sdf_net(sd, x[:M].double(), "sdf_net.")


def run(prec, split):
    net.sdf_split = split
    s = net.sdf(xc, prec).reshape(-1)
    sdf2, grad, c = net.fwd_stash(points_struct(x=xc), N, prec)
    StashCache.release(c["lease"])
    torch.cuda.synchronize()
    t = []
    for fn in (lambda: net.sdf(xc, prec), lambda: StashCache.release(net.fwd_stash(points_struct(x=xc), N, prec)[2]["lease"])):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t.append((time.perf_counter() - t0) / 10 * 1e3)
    e1 = float((s[:M].cpu().double() - ref_sdf).abs().max())
    e2 = float((sdf2[:M].cpu().double() - ref_sdf).abs().max())
    eg = float((grad[:M].cpu().double() - ref_grad).abs().max())
    return e1, e2, eg, t


for name, prec, split in (("f32", nw.PREC_F32, False), ("f16 plain", nw.PREC_F16, False), ("f16 split", nw.PREC_F16, True),
                          ("bf16", nw.PREC_BF16, False)):
    e1, e2, eg, t = run(prec, split)
    print("%-10s max|sdf - fp64|: infer %.2e  fwd %.2e   max|grad - fp64| %.2e   ms per 131072 points: infer %.3f  fwd %.3f"
          % (name, e1, e2, eg, t[0], t[1]), flush=True)
