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
t = out.reshape(-1)[:1024].cpu().view(8, 128)
for w in (0, 4, 3, 7):
    after = t[w, 0:34].tolist(); before = t[w, 64:64+33].tolist()
    print("wave %d: start %.0f end %.0f" % (w, after[0], t[w, 127]))
    print("  phase durations (barrier to barrier):", " ".join("%d" % (after[i+1]-after[i]) for i in range(33)))
    print("  own work in phase:                   ", " ".join("%d" % (before[i]-after[i]) for i in range(33)))
