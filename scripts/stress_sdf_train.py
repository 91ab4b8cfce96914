"""Determinism stress: repeats the SDF training path and reports which stage (if any) is flaky."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import neuralrecon_w_amd as nw
from neuralrecon_w_amd.neuconw import points_struct
from neuralrecon_w_amd.stash import WgradBatch
from tests.test_gpu_sdf import _mk

prec = nw.PREC_BF16
W, n = 256, 777
net = _mk(W, 8, (4,), seed=3)
g = torch.Generator().manual_seed(9)
x = ((torch.rand(n, 3, generator=g) * 2 - 1) * 0.9).cuda()
w_sdf = torch.randn(n, generator=g).cuda(); w_grad = torch.randn(n, 3, generator=g).cuda()
w_feat = (torch.randn(n, W, generator=g) * 0.1).cuda()
ref = None
bad = {"fwd": 0, "bwd": 0, "wgrad": 0}
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    pts = points_struct(x=x)
    sdf, grad, ctx = net.fwd_stash(pts, n, prec)
    ar, ids = ctx["arena"], ctx["ids"]
    ar.from_rows(ids["dfeat"], w_feat)
    fwd_snap = torch.cat([sdf, grad.reshape(-1), ar.to_rows(ids["feat"], W).reshape(-1)] +
                         [ar.to_rows(ids["t"][l], W).reshape(-1) for l in range(8)])
    net.bwd_stash(ctx, w_sdf, w_grad)
    bwd_snap = torch.cat([ar.to_rows(ids["zbar"][l], W).reshape(-1) for l in range(8)] +
                         [ar.to_rows(ids["qbar"][l], W if l else 39).reshape(-1) for l in range(9)])
    plan = ctx["plan"]
    plan.g_arena.zero_()
    b = WgradBatch(x.device, prec, n); net.add_wgrads(ctx, b); b.run()
    dense = plan.g_arena.clone()
    torch.cuda.synchronize()
    if ref is None:
        ref = (fwd_snap, bwd_snap, dense)
        continue
    if not torch.equal(fwd_snap, ref[0]): bad["fwd"] += 1
    if not torch.equal(bwd_snap, ref[1]): bad["bwd"] += 1
    e = float((dense - ref[2]).abs().max() / ref[2].abs().max())
    if e > 1e-3: bad["wgrad"] += 1; print("iter", it, "wgrad rel diff", e)
print("flaky counts:", bad)
