import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import neuralrecon_w_amd as nw
from tests._build import build_system, loss_from_outputs
from tests._util import synth_rays
emb, neuconw, nerf, rdr = build_system(seed=5, prec=nw.PREC_F32)
rays, ts, label, rgbs = [t.cuda() for t in synth_rays(64, seed=11, n_vocab=64)]
bg = torch.zeros(1, 3, device="cuda")
train = nw.TrainStep(rdr, [emb, neuconw, nerf], loss_from_outputs, lr=1e-3, eps=1e-7, clip=0.99)
print(type(train.opt), train.opt.defaults.get("fused"))
w = neuconw.sdf_net.lin3.weight_v if hasattr(neuconw.sdf_net, "lin3") else list(neuconw.sdf_net.parameters())[5]
for i in range(3):
    w0 = w.detach().clone(); v0 = train.fp.flat._version
    loss, _ = train(rays, ts, label, rgbs, background_rgb=bg, cos_anneal_ratio=0.0, perturb_overwrite=0)
    print(i, float(loss), "flat ver", v0, train.fp.flat._version, "dw", float((w - w0).abs().max()),
          "gnorm", float(train.fp.flat_grad.norm()), "w.grad norm", float(w.grad.norm()),
          "pver", neuconw.sdf_net._param_version()[-3:])
