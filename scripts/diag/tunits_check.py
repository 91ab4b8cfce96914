"""GPU: the value-only SDF kernels (t-units, round 6) against the training forward's value chain (z-units) and the fp64 oracle on the
SAME points: any difference beyond rounding noise is a bug of the unit change.  W = 256 / 512; bf16, fp16 plain, fp16 split."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts", "diag"))
import neuralrecon_w_amd as nw  # noqa: E402
from neuralrecon_w_amd.neuconw import points_struct  # noqa: E402
from neuralrecon_w_amd.stash import StashCache  # noqa: E402
from oracle import neuconw_oracle as O  # noqa: E402
from sdf_infer_units import build, points  # noqa: E402

for W in (256, 512):
    net = build(W)
    sd = {"sdf_net." + k: v.detach().cpu().double() for k, v in net.state_dict().items()}
    for N in (131072, 8192 + 37):
        x = points(N)
        xc = x.cuda()
        M = min(N, 16384)
        ref = O.sdf_net(sd, x[:M].double(), "sdf_net.", with_grad=False)[0]
        for name, prec, split in (("bf16 plain", nw.PREC_BF16, False), ("f16 plain", nw.PREC_F16, False), ("f16 split", nw.PREC_F16, True)):
            net.sdf_split = split
            s_inf = net.sdf(xc, prec).reshape(-1)[:M].cpu().double()
            s_fwd, _, c = net.fwd_stash(points_struct(x=xc), N, prec)
            StashCache.release(c["lease"])
            s_fwd = s_fwd[:M].cpu().double()
            e_inf, e_fwd = (s_inf - ref).abs(), (s_fwd - ref).abs()
            w = int(e_inf.argmax())
            print("W=%d N=%d %-10s |infer - fp64| max %.2e mean %.2e   |fwd - fp64| max %.2e mean %.2e   |infer - fwd| max %.2e   worst point %d: |x| %.3f sdf %.4f"
                  % (W, N, name, float(e_inf.max()), float(e_inf.mean()), float(e_fwd.max()), float(e_fwd.mean()), float((s_inf - s_fwd).abs().max()),
                     w, float(x[w].norm()), float(ref[w])), flush=True)
