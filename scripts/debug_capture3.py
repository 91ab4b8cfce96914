import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import neuralrecon_w_amd as nw
from tests._build import build_system, loss_from_outputs
from tests._util import synth_rays
mode = sys.argv[1]
rays, ts, label, rgbs = [t.cuda() for t in synth_rays(64, seed=12, n_vocab=64)]
bg = torch.zeros(1, 3, device="cuda")
emb, neuconw, nerf, rdr = build_system(seed=6, prec=nw.PREC_F32)
rdr.sync_free = True
train = nw.TrainStep(rdr, [emb, neuconw, nerf], loss_from_outputs, lr=1e-3, eps=1e-7, clip=0.99,
                     capture=True, capture_warmup=3)
for i in range(3):
    loss, out = train.eager_step(rays, ts, label, rgbs, background_rgb=bg, cos_anneal_ratio=0.15 * i, perturb_overwrite=0)
    if "nofloat" not in mode:
        print(" step", i, float(loss), flush=True)
if "olddetached" in mode:
    loss = loss.detach(); out = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}
    import gc; gc.collect()
elif "oldloss" in mode:
    del out
elif "oldout" in mode:
    del loss
elif "oldalive" not in mode:
    del loss, out
if "clone" in mode:
    r2, t2, l2, g2, b2 = rays.clone(), ts.clone(), label.clone(), rgbs.clone(), bg.clone()
else:
    r2, t2, l2, g2, b2 = rays, ts, label, rgbs, bg
cos = torch.zeros(1, device="cuda")
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    res = train._fwd_bwd(r2, t2, l2, g2, b2, cos, dict(perturb_overwrite=0))
    if "keep" not in mode:
        del res
    train._update()
print("captured", mode, flush=True)
g.replay(); torch.cuda.synchronize()
print("replayed", mode)
