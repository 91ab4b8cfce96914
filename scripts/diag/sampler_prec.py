"""Which part of the fp16 mode's output error comes from the SAMPLER's SDF queries (they place the samples) and which
from the training passes at those samples?  Renders the test_gpu_fullsize.py cases with the sampler in fp32 / fp16 and
the MLP passes in fp16 / fp32 and prints the output errors against the all-fp32 render (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import neuralrecon_w_amd as nw
from tests._build import build_system
from tests._util import rel_err, synth_rays
from tests.test_gpu_fullsize import _jitter

for (ns, ni, R) in ((16, 16, 40), (64, 64, 64)):
    outs = {}
    for name, prec, sp in (("all f32", nw.PREC_F32, None), ("all f16", nw.PREC_F16, None), ("f16, sampler f32", nw.PREC_F16, nw.PREC_F32),
                           ("f32, sampler f16", nw.PREC_F32, nw.PREC_F16), ("all bf16", nw.PREC_BF16, None),
                           ("bf16, sampler f32", nw.PREC_BF16, nw.PREC_F32)):
        emb, neuconw, nerf, rdr = build_system(W=256, n_a=48, n_vocab=100, nerf_w=256, color_hidden=256, head=128, seed=5,
                                               prec=prec, n_samples=ns, n_importance=ni)
        _jitter(neuconw)
        rdr.sampler_prec = sp
        rays, ts, label, rgbs = synth_rays(R, 77, 100)
        with torch.no_grad():
            out = rdr.render(rays.cuda(), ts.cuda(), label.cuda(), perturb_overwrite=0, background_rgb=torch.zeros(1, 3).cuda(),
                             cos_anneal_ratio=0.3)
        outs[name] = {k: out[k].detach().float().cpu() for k in ("color", "depth", "weights_sum", "gradient_error")}
    ref = outs["all f32"]
    for name, o in outs.items():
        if name != "all f32":
            print("%d+%d %-20s" % (ns, ni, name), {k: "%.2e" % rel_err(o[k], ref[k]) for k in o})
