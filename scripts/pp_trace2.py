import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import neuralrecon_w_amd as nw
dev = torch.device("cuda:0")
net = nw.SDFNetwork(d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=(4,)).to(dev)
x = (torch.rand(131072, 3, device=dev) * 2 - 1)
for _ in range(3): out = net.sdf(x, prec=nw.PREC_BF16)
torch.cuda.synchronize()
t = out.reshape(-1)[:8 * 512].cpu().view(8, 64, 8)
for seg in (6, 7, 8, 9):
    for w in (0, 4, 1, 5):
        r = t[w, seg, :5].tolist()
        print("segment %d wave %d: start %6.0f  steps 0-4 %5.0f  4-8 %5.0f  8-12 %5.0f  12-16 %5.0f   total %5.0f" % (seg, w, r[0], r[1]-r[0], r[2]-r[1], r[3]-r[2], r[4]-r[3], r[4]-r[0]))
    print("   next segment starts (wave 0) at %6.0f -> gap after wave0's end %5.0f" % (t[0, seg + 1, 0], t[0, seg + 1, 0] - t[0, seg, 4]))
