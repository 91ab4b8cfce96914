import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import neuralrecon_w_amd as nw
from tests._build import build_system, loss_from_outputs
from tests._util import synth_rays
mode = sys.argv[1]
rays, ts, label, rgbs = [t.cuda() for t in synth_rays(64, seed=12, n_vocab=64)]
bg = torch.zeros(1, 3, device="cuda")
def run(capture, steps=7, keep=True):
    emb, neuconw, nerf, rdr = build_system(seed=6, prec=nw.PREC_F32)
    rdr.sync_free = True
    train = nw.TrainStep(rdr, [emb, neuconw, nerf], loss_from_outputs, lr=1e-3, eps=1e-7, clip=0.99,
                         capture=capture, capture_warmup=3)
    if mode == "noclone" and capture:
        train._clone = False
    for i in range(steps):
        loss, out = train(rays, ts, label, rgbs, background_rgb=bg, cos_anneal_ratio=0.15 * i, perturb_overwrite=0)
        print(" step", i, float(loss), flush=True)
    return emb, neuconw, nerf, rdr, train
if mode in ("full",):
    a = run(False)
b = run(True)
print("done", mode)
