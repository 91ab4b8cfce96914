import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import neuralrecon_w_amd as nw
from oracle import neuconw_oracle as O
from tests._build import build_system, loss_from_outputs, named_params, state_dict_cpu
from tests._util import rel_err, synth_rays
from tests.test_gpu_fullsize import CFG, _jitter
W, ns, ni = 512, 8, 16
emb, neuconw, nerf, rdr = build_system(W=W, n_a=48, n_vocab=100, nerf_w=256, color_hidden=256, head=128, seed=5, prec=nw.PREC_F32, n_samples=ns, n_importance=ni)
_jitter(neuconw)
R = 40
rays, ts, label, rgbs = synth_rays(R, 77, 100)
out = rdr.render(rays.cuda(), ts.cuda(), label.cuda(), perturb_overwrite=0, background_rgb=torch.zeros(1, 3).cuda(), cos_anneal_ratio=0.3)
loss = loss_from_outputs(out, rgbs.cuda()); loss.backward()
params = named_params(emb, neuconw, nerf)
res = {}
for dt in (torch.float32, torch.float64):
    sd = state_dict_cpu(emb, neuconw, nerf, dt); sd = {k: v.requires_grad_(True) for k, v in sd.items()}
    cfg = dict(CFG, n_samples=ns, n_importance=ni)
    ref = O.render(sd, cfg, rays.to(dt), ts, label, 0.3, torch.zeros(1, 3, dtype=dt))
    l = O.neuconw_loss(ref, rgbs.to(dt), cfg)
    names = list(sd)
    res[dt] = (ref, dict(zip(names, torch.autograd.grad(l, [sd[k] for k in names], allow_unused=True))))
    print(dt, "z diff vs gpu weights:", rel_err(out["weights"].detach().cpu(), ref["weights"]))
g32, g64 = res[torch.float32][1], res[torch.float64][1]
for k in g64:
    if g64[k] is None: continue
    e_gpu = rel_err(params[k].grad.cpu(), g64[k]); e_cpu = rel_err(g32[k], g64[k])
    if e_gpu > 1e-3 or e_cpu > 1e-3:
        print("%-50s gpu-vs-fp64 %.2e   cpu32-vs-fp64 %.2e   max|g| %.2e" % (k, e_gpu, e_cpu, float(g64[k].abs().max())))
print("done")
