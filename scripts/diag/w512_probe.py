"""GPU, probe libraries: the W = 512 split kernels (value-only infer and the training forward at the shipped batch) under build variants
(NEUCONW_HIP_LIB): time per launch by HIP events, median of 20.   python scripts/diag/w512_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts", "diag"))
import neuralrecon_w_amd as nw  # noqa: E402
from neuralrecon_w_amd.neuconw import points_struct  # noqa: E402
from neuralrecon_w_amd.stash import StashCache  # noqa: E402
from sdf_infer_units import build, points, timed  # noqa: E402

net = build(512)
out = []
for N in (49152, 1 << 20):
    xc = points(N).cuda()
    net.sdf_split = True
    t_inf = timed(lambda: net.sdf(xc, nw.PREC_F16))
    out.append("infer split %8d pts %.4f ms (best %.4f)" % (N, t_inf[0], t_inf[1]))
    if N < 100000:
        for adj in (0, 1, 2):
            net.adj_split = adj
            t = timed(lambda: StashCache.release(net.fwd_stash(points_struct(x=xc), N, nw.PREC_F16)[2]["lease"]))
            out.append("sdf_fwd adj=%d %8d pts %.4f ms (best %.4f)" % (adj, N, t[0], t[1]))
        net.sdf_split = False
        t = timed(lambda: net.sdf(xc, nw.PREC_F16))
        out.append("infer plain %8d pts %.4f ms (best %.4f)" % (N, t[0], t[1]))
print(os.environ.get("NEUCONW_HIP_LIB", "product").split("/")[-1], "|", " | ".join(out), flush=True)
