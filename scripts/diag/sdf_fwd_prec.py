"""sdf / gradient / feature error of the 16-bit sdf_fwd against the fp32 mode, per width (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import neuralrecon_w_amd as nw
from neuralrecon_w_amd.neuconw import points_struct
from tests.test_gpu_sdf import _mk
from tests._util import rel_err

for W in (64, 256, 512):
    net = _mk(W, 8, (4,), seed=3)
    g = torch.Generator().manual_seed(9)
    x = ((torch.rand(4096, 3, generator=g) * 2 - 1) * 0.9).cuda()
    ref = None
    for name, prec in (("f32", nw.PREC_F32), ("bf16", nw.PREC_BF16), ("f16", nw.PREC_F16)):
        sdf, grad, ctx = net.fwd_stash(points_struct(x=x), x.shape[0], prec)
        feat = ctx["arena"].to_rows(ctx["ids"]["feat"], W)
        cur = (sdf.clone(), grad.clone(), feat.clone())
        if ref is None:
            ref = cur
            continue
        print("W=%d %s: sdf %.2e grad %.2e feat %.2e   |grad| mean err %.2e" % (
            W, name, rel_err(cur[0].cpu(), ref[0].cpu()), rel_err(cur[1].cpu(), ref[1].cpu()), rel_err(cur[2].cpu(), ref[2].cpu()),
            float((cur[1].norm(dim=-1) - ref[1].norm(dim=-1)).abs().mean())))
