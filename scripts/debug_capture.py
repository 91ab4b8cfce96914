import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import neuralrecon_w_amd as nw
from tests._build import build_system, loss_from_outputs
from tests._util import synth_rays
stage = sys.argv[1]
emb, neuconw, nerf, rdr = build_system(seed=6, prec=nw.PREC_F32)
rdr.sync_free = True
rays, ts, label, rgbs = [t.cuda() for t in synth_rays(64, seed=12, n_vocab=64)]
bg = torch.zeros(1, 3, device="cuda")
train = nw.TrainStep(rdr, [emb, neuconw, nerf], loss_from_outputs, lr=1e-3, capture=False)
train.opt = torch.optim.Adam([train.fp.flat], lr=1e-3, eps=1e-7, fused=True, capturable=True)
for i in range(3):
    train.eager_step(rays, ts, label, rgbs, background_rgb=bg, cos_anneal_ratio=0.0, perturb_overwrite=0)
cos = torch.zeros(1, device="cuda")
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    if stage == "two":
        train._fwd_bwd(rays, ts, label, rgbs, bg, cos, dict(perturb_overwrite=0))
    elif stage == "zero":
        train.fp.zero_grad()
    elif stage == "sampler":
        with torch.no_grad():
            o = rdr.sdf(torch.rand(4096, 3, device="cuda"))
    elif stage == "fwd":
        with torch.no_grad():
            out = rdr.render(rays, ts, label, background_rgb=bg, cos_anneal_ratio=cos, perturb_overwrite=0)
    elif stage == "fwdloss":
        out = rdr.render(rays, ts, label, background_rgb=bg, cos_anneal_ratio=cos, perturb_overwrite=0)
        loss = loss_from_outputs(out, rgbs)
    elif stage == "bwd":
        train._fwd_bwd(rays, ts, label, rgbs, bg, cos, dict(perturb_overwrite=0))
    elif stage == "update":
        train._update()
    elif stage == "clip":
        torch.nn.utils.clip_grad_norm_([train.fp.flat], 0.99)
    elif stage == "adam":
        train.opt.step()
    elif stage == "all":
        train._fwd_bwd(rays, ts, label, rgbs, bg, cos, dict(perturb_overwrite=0))
        train._update()
    elif stage == "bwdclip":
        train._fwd_bwd(rays, ts, label, rgbs, bg, cos, dict(perturb_overwrite=0))
        torch.nn.utils.clip_grad_norm_([train.fp.flat], 0.99)
    elif stage == "bwdadam":
        train._fwd_bwd(rays, ts, label, rgbs, bg, cos, dict(perturb_overwrite=0))
        train.opt.step()
print("captured", stage)
if stage == "two":
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, pool=g.pool()):
        train._update()
    print("captured second")
g.replay(); torch.cuda.synchronize()
print("replayed", stage)
